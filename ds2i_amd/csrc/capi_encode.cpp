// Host side of the GPU index encoder (include/ds2i_hip.h: ds2i_hip_encode_index): uploads the
// postings, runs the plan pass (findBestB per block part, sizes), lays the lists out from the sizes
// (block_posting_list::write's layout, block_posting_list.hpp:13-53), runs the write pass, and freezes the
// block_freq_index image around the device-written list bytes (block_freq_index.hpp:18-70, 124-134).
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <vector>

#include "capi_blob.hpp"
#include "capi_internal.hpp"
#include "host_index.hpp"

extern "C" {
size_t ds2i_sizeof_enc_args();
hipError_t ds2i_launch_encode(int write, const void* args, unsigned grid, hipStream_t s);
}

namespace {
struct EncArgsHost { // mirrors EncArgs in encode_kernels.hip
    const uint32_t* docs;
    const uint32_t* freqs;
    const uint64_t* list_in;
    const uint32_t* blk_list;
    const uint32_t* list_blk0;
    uint32_t nblocks;
    uint8_t* bsel;
    uint32_t* psize;
    uint32_t* bmax;
    const uint64_t* blk_out;
    const uint64_t* list_out;
    uint8_t* out;
};
struct DevFree {
    std::vector<void*> p;
    ~DevFree() { for (void* x : p) if (x) (void)hipFree(x); }
    template <class T> hipError_t alloc(T** out, size_t bytes) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
        if (e == hipSuccess) p.push_back(q);
        *out = (T*)q;
        return e;
    }
};
} // namespace

extern "C" int ds2i_hip_encode_index(int device, int index_kind, uint64_t num_docs, uint64_t nlists, const uint64_t* list_offsets,
                                     const uint32_t* docs, const uint32_t* freqs, ds2i_blob** image, double* device_ms) {
    if (!list_offsets || !docs || !freqs || !image) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_encode_index: null argument");
    if (index_kind != DS2I_BLOCK_OPTPFOR) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_encode_index: the GPU encoder writes block_optpfor indexes");
    if (sizeof(EncArgsHost) != ds2i_sizeof_enc_args()) return ds2i_set_error(DS2I_EINVAL, "EncArgs layout mismatch");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return ds2i_set_error(DS2I_EDEVICE, "ds2i_hip_encode_index: no such HIP device");
    const uint64_t total = list_offsets[nlists];
    std::vector<uint32_t> blk_list, list_blk0(nlists);
    uint64_t nblocks = 0;
    for (uint64_t t = 0; t < nlists; ++t) {
        if (list_offsets[t + 1] <= list_offsets[t]) return ds2i_set_error(DS2I_EINVAL, "List must be nonempty"); // block_freq_index.hpp:31
        const uint64_t n = list_offsets[t + 1] - list_offsets[t];
        if (n > 0xFFFFFFFFull) return ds2i_set_error(DS2I_EINVAL, "posting list longer than 2^32");
        list_blk0[t] = (uint32_t)nblocks;
        nblocks += (n + 127) / 128;
    }
    if (nblocks >= (1ull << 32)) return ds2i_set_error(DS2I_EINVAL, "more than 2^32 blocks");
    try {
        blk_list.resize(nblocks);
    } catch (std::bad_alloc const&) {
        return ds2i_set_error(DS2I_ENOMEM, "out of host memory");
    }
    for (uint64_t t = 0; t < nlists; ++t) {
        const uint64_t nb = (list_offsets[t + 1] - list_offsets[t] + 127) / 128;
        std::fill(blk_list.begin() + list_blk0[t], blk_list.begin() + list_blk0[t] + nb, (uint32_t)t);
    }
    HIP_OK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    const unsigned grid = (unsigned)std::min<uint64_t>(nblocks ? nblocks : 1, (uint64_t)(prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * 32);
    DevFree dev;
    uint32_t *d_docs, *d_freqs, *d_blk_list, *d_list_blk0, *d_psize, *d_bmax;
    uint64_t *d_list_in, *d_blk_out, *d_list_out;
    uint8_t *d_bsel, *d_out = nullptr;
    HIP_OK(dev.alloc(&d_docs, 4 * total));
    HIP_OK(dev.alloc(&d_freqs, 4 * total));
    HIP_OK(dev.alloc(&d_list_in, 8 * (nlists + 1)));
    HIP_OK(dev.alloc(&d_blk_list, 4 * nblocks));
    HIP_OK(dev.alloc(&d_list_blk0, 4 * nlists));
    HIP_OK(dev.alloc(&d_bsel, 2 * nblocks));
    HIP_OK(dev.alloc(&d_psize, 8 * nblocks));
    HIP_OK(dev.alloc(&d_bmax, 4 * nblocks));
    HIP_OK(dev.alloc(&d_blk_out, 8 * (nblocks + 1)));
    HIP_OK(dev.alloc(&d_list_out, 8 * nlists));
    HIP_OK(hipMemcpy(d_docs, docs, 4 * total, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_freqs, freqs, 4 * total, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_list_in, list_offsets, 8 * (nlists + 1), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_blk_list, blk_list.data(), 4 * nblocks, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_list_blk0, list_blk0.data(), 4 * nlists, hipMemcpyHostToDevice));
    struct Events { // destroyed on every path out of the function
        hipEvent_t e[4] = {};
        ~Events() { for (auto x : e) if (x) (void)hipEventDestroy(x); }
    } evs;
    for (auto& x : evs.e) HIP_OK(hipEventCreate(&x));
    const hipEvent_t e0 = evs.e[0], e1 = evs.e[1], e2 = evs.e[2], e3 = evs.e[3];
    EncArgsHost a{};
    a.docs = d_docs;
    a.freqs = d_freqs;
    a.list_in = d_list_in;
    a.blk_list = d_blk_list;
    a.list_blk0 = d_list_blk0;
    a.nblocks = (uint32_t)nblocks;
    a.bsel = d_bsel;
    a.psize = d_psize;
    a.bmax = d_bmax;
    // ---- plan pass
    HIP_OK(hipEventRecord(e0, nullptr));
    if (nblocks) HIP_OK(ds2i_launch_encode(0, &a, grid, nullptr));
    HIP_OK(hipEventRecord(e1, nullptr));
    std::vector<uint32_t> psize(2 * nblocks);
    HIP_OK(hipMemcpy(psize.data(), d_psize, 8 * nblocks, hipMemcpyDeviceToHost));
    // ---- layout: vbyte(n) | block_max[nb] | block_endpoint[nb - 1] | blocks (docs part, freqs part each)
    std::vector<uint64_t> blk_out(nblocks + 1), list_out(nlists), list_end(nlists);
    uint64_t cursor = 0;
    for (uint64_t t = 0; t < nlists; ++t) {
        const uint64_t n = list_offsets[t + 1] - list_offsets[t], nb = (n + 127) / 128;
        const uint32_t vl = 1u + (n >= (1u << 7)) + (n >= (1u << 14)) + (n >= (1u << 21)) + (n >= (1u << 28));
        list_out[t] = cursor;
        cursor += vl + 8 * nb - 4;
        for (uint64_t b = 0; b < nb; ++b) {
            const uint64_t g = list_blk0[t] + b;
            blk_out[g] = cursor;
            cursor += (uint64_t)psize[2 * g] + psize[2 * g + 1];
        }
        list_end[t] = cursor;
    }
    blk_out[nblocks] = cursor;
    HIP_OK(dev.alloc(&d_out, cursor + 64));
    HIP_OK(hipMemcpy(d_blk_out, blk_out.data(), 8 * (nblocks + 1), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_list_out, list_out.data(), 8 * nlists, hipMemcpyHostToDevice));
    a.blk_out = d_blk_out;
    a.list_out = d_list_out;
    a.out = d_out;
    // ---- write pass
    HIP_OK(hipEventRecord(e2, nullptr));
    if (nblocks) HIP_OK(ds2i_launch_encode(1, &a, grid, nullptr));
    HIP_OK(hipEventRecord(e3, nullptr));
    HIP_OK(hipEventSynchronize(e3));
    float ms_plan = 0.f, ms_write = 0.f;
    HIP_OK(hipEventElapsedTime(&ms_plan, e0, e1));
    HIP_OK(hipEventElapsedTime(&ms_write, e2, e3));
    if (device_ms) *device_ms = (double)ms_plan + ms_write;
    try {
        ds2i_host::bytes_t lists(cursor);
        HIP_OK(hipMemcpy(lists.data(), d_out, cursor, hipMemcpyDeviceToHost));
        ds2i_host::block_index_builder builder(ds2i_host::CODEC_OPTPFOR, num_docs);
        builder.set_encoded_lists(std::move(lists), list_end);
        std::unique_ptr<ds2i_blob> blob(new ds2i_blob);
        builder.freeze(blob->data);
        *image = blob.release();
    } catch (std::bad_alloc const&) {
        return ds2i_set_error(DS2I_ENOMEM, "out of host memory");
    } catch (std::exception const& e) {
        return ds2i_set_error(DS2I_EFORMAT, e.what());
    }
    return DS2I_OK;
}

// The synthetic collection (ds2i_build.h: ds2i_synth_params) generated on the host threads and encoded ON THE GPU:
// the fast index-construction path of the benchmark loop. Produces the same two images as ds2i_synth_build(...,
// DS2I_BLOCK_OPTPFOR, ...), byte for byte.
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include "../../include/ds2i_build.h"
#include "host_synth.hpp"

extern "C" int ds2i_hip_synth_encode(int device, const ds2i_synth_params* pp, int threads, ds2i_blob** index_image,
                                     ds2i_blob** wand_image, uint64_t* total_postings, double* generate_s, double* device_ms) {
    if (!pp || !index_image) return ds2i_set_error(DS2I_EINVAL, "ds2i_hip_synth_encode: null argument");
    using namespace ds2i_host;
    try {
        synth_params p;
        p.seed = pp->seed; p.num_docs = pp->num_docs; p.num_terms = pp->num_terms; p.zipf_exp = pp->zipf_exp;
        p.top_df_frac = pp->top_df_frac; p.min_len = pp->min_len; p.clustered_every = pp->clustered_every;
        p.topics = pp->topics; p.topic_boost = pp->topic_boost;
        if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> sizes;
        synth_doc_sizes(p, sizes);
        std::vector<float> norm_lens;
        compute_norm_lens(sizes.data(), p.num_docs, norm_lens);
        std::vector<uint32_t>().swap(sizes);
        const uint32_t V = p.num_terms;
        std::vector<std::vector<uint32_t>> ld(V), lf(V);
        std::vector<float> max_w(V);
        std::atomic<uint32_t> next(0);
        std::string err;
        std::mutex err_mu;
        auto run_pool = [&](auto fn) {
            std::vector<std::thread> pool;
            for (int i = 0; i < threads; ++i) pool.emplace_back(fn);
            for (auto& th : pool) th.join();
        };
        run_pool([&]() {
            try {
                for (;;) {
                    const uint32_t t = next.fetch_add(1);
                    if (t >= V) break;
                    const uint64_t n = synth_list(p, t, ld[t], lf[t]);
                    ld[t].resize(n);
                    lf[t].resize(n);
                    ld[t].shrink_to_fit();
                    lf[t].shrink_to_fit();
                    max_w[t] = list_max_weight(norm_lens.data(), n, ld[t].data(), lf[t].data());
                }
            } catch (std::exception const& e) {
                std::lock_guard<std::mutex> g(err_mu);
                err = e.what();
            }
        });
        if (!err.empty()) return ds2i_set_error(DS2I_EFORMAT, err.c_str());
        std::vector<uint64_t> offs(V + 1, 0);
        for (uint32_t t = 0; t < V; ++t) offs[t + 1] = offs[t] + ld[t].size();
        std::vector<uint32_t> docs(offs[V] ? offs[V] : 1), freqs(offs[V] ? offs[V] : 1);
        next = 0;
        run_pool([&]() {
            for (;;) {
                const uint32_t t = next.fetch_add(1);
                if (t >= V) break;
                std::memcpy(docs.data() + offs[t], ld[t].data(), 4 * ld[t].size());
                std::memcpy(freqs.data() + offs[t], lf[t].data(), 4 * lf[t].size());
                std::vector<uint32_t>().swap(ld[t]);
                std::vector<uint32_t>().swap(lf[t]);
            }
        });
        if (generate_s) *generate_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (total_postings) *total_postings = offs[V];
        int rc = ds2i_hip_encode_index(device, DS2I_BLOCK_OPTPFOR, p.num_docs, V, offs.data(), docs.data(), freqs.data(), index_image, device_ms);
        if (rc) return rc;
        if (wand_image) {
            std::unique_ptr<ds2i_blob> wb(new ds2i_blob);
            wand_freeze(norm_lens, max_w, wb->data);
            *wand_image = wb.release();
        }
    } catch (std::bad_alloc const&) {
        return ds2i_set_error(DS2I_ENOMEM, "out of host memory");
    } catch (std::exception const& e) {
        return ds2i_set_error(DS2I_EFORMAT, e.what());
    }
    return DS2I_OK;
}
