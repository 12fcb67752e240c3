// HIP kernels (gfx950, wave64) that ENCODE block_optpfor posting lists: the build-side counterpart of the decoders
// (SURVEY.md §8(f) item 2). One wavefront per 128-posting block; two passes over the same code:
//   plan   findBestB for the docs part and the freqs part of every block (block_codecs.hpp:156-182: every b of
//          OPTPFor's possLogs that the early-stop rule admits is tried, tryB = ceil(128*b/32) words + the Simple16 word
//          count of that b's exceptions, the LAST minimum wins), sizes of the interpolative tail blocks, block_max
//   write  the bytes, at the offsets the host derived from the plan (block_posting_list::write,
//          block_posting_list.hpp:13-53: vbyte(n) | block_max[] | block_endpoint[] | docs part, freqs part per block)
// The output is byte-identical to the host encoder (host_encode.hpp / host_index.hpp::write_posting_list) -- that is the
// parity contract, tested in tests/test_gpu.py::test_gpu_encode_is_byte_identical.
#include <hip/hip_runtime.h>

#include "device_codecs.hpp"

using namespace ds2i_dev;

namespace {

struct EncArgs {
    const uint32_t* docs;      // postings of all lists, concatenated
    const uint32_t* freqs;
    const uint64_t* list_in;   // nlists + 1 posting offsets
    const uint32_t* blk_list;  // per block: its list
    const uint32_t* list_blk0; // per list: its first block (global numbering)
    uint32_t nblocks;
    uint8_t* bsel;             // 2 per block: chosen b of the docs / freqs part (full blocks)
    uint32_t* psize;           // 2 per block: bytes of the docs / freqs part
    uint32_t* bmax;            // per block: last doc-id
    const uint64_t* blk_out;   // write pass: byte offset of the block's bytes in `out` (nblocks + 1 entries)
    const uint64_t* list_out;  // write pass: byte offset of the list (its vbyte(n)) in `out`
    uint8_t* out;
};

__device__ static const uint8_t ENC_LOGS[17] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 16, 20, 32};
// Simple16 layouts as (count, width) runs -- the same table the decoder uses (device_codecs.hpp S16_DESC)

struct EncLds {
    uint32_t v[2][128];      // gap-1 / freq-1 of the block, index order
    uint32_t exc[256];       // exception array of the b being tried: nExc position deltas, then nExc high parts - 1
    uint8_t len[256];        // bit length of exc[i]
    uint16_t acc[28][30];    // acc[j][l]: bit s set iff a value of bit length l may sit in field j of selector s
    uint8_t s16_n[16];       // fields per selector
    uint8_t s16_shift[16][28]; // left shift of field j inside the 28 payload bits
    uint32_t outw[1 + 256 + 128 + 8]; // the part being written, as dwords
};

DS2I_DEV uint32_t wave_and_all(uint32_t x) {
    x &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x111, 0xF, 0xF, false);
    x &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x112, 0xF, 0xF, false);
    x &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xF, 0xF, false);
    x &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xF, 0xF, false);
    x &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x142, 0xA, 0xF, false);
    x &= (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x143, 0xC, 0xF, false);
    return bcast(x, 63);
}
DS2I_DEV uint32_t wave_or_all(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return bcast(x, 63);
}

DS2I_DEV void enc_tables_init(EncLds& L) {
    const uint32_t lane = lane_id();
    if (lane < 16) {
        const uint32_t d = S16_DESC[lane];
        const uint32_t c0 = d & 31, w0 = (d >> 5) & 31, c1 = (d >> 10) & 31, w1 = (d >> 15) & 31, c2 = (d >> 20) & 31, w2 = d >> 25;
        L.s16_n[lane] = (uint8_t)(c0 + c1 + c2);
        uint32_t end = 0;
        for (uint32_t j = 0; j < 28; ++j) {
            uint32_t w = j < c0 ? w0 : j < c0 + c1 ? w1 : j < c0 + c1 + c2 ? w2 : 0;
            end += w;
            L.s16_shift[lane][j] = (uint8_t)(w ? 28 - end : 0);
        }
    }
    for (uint32_t e = lane; e < 28 * 30; e += 64) {
        const uint32_t j = e / 30, l = e % 30;
        uint32_t m = 0;
        for (uint32_t s = 0; s < 16; ++s) {
            const uint32_t d = S16_DESC[s];
            const uint32_t c0 = d & 31, w0 = (d >> 5) & 31, c1 = (d >> 10) & 31, w1 = (d >> 15) & 31, c2 = (d >> 20) & 31, w2 = d >> 25;
            const uint32_t n = c0 + c1 + c2;
            const uint32_t w = j < c0 ? w0 : j < c0 + c1 ? w1 : w2;
            if (j >= n || l <= w) m |= 1u << s;
        }
        L.acc[j][l] = (uint16_t)m;
    }
    wave_sync();
}

// Simple16 of exc[0 .. need): FastPFor's greedy -- at every position the first selector (0..15) whose fields hold the
// next values wins; a short tail only has to fit the fields it uses. Returns the number of words; with EMIT they are
// stored to `dst` (LDS).
template <bool EMIT>
DS2I_DEV uint32_t simple16_words(EncLds& L, uint32_t need, uint32_t* dst) {
    const uint32_t lane = lane_id();
    for (uint32_t i = lane; i < need; i += 64) {
        const uint32_t x = L.exc[i];
        const uint32_t l = x ? 32u - (uint32_t)__builtin_clz(x) : 0u;
        L.len[i] = (uint8_t)(l > 29 ? 29 : l);
    }
    wave_sync();
    uint32_t i = 0, words = 0;
    while (i < need) {
        const uint32_t rem = need - i;
        uint32_t m = 0xFFFFu;
        if (lane < 28) m = L.acc[lane][lane < rem ? L.len[i + lane] : 0];
        const uint32_t all = wave_and_all(m);
        const uint32_t sel = (uint32_t)__builtin_ctz(all | 0x10000u); // `all` is never 0: selector 15 holds any 28-bit value
        const uint32_t ns = L.s16_n[sel < 16 ? sel : 15];
        const uint32_t cnt = rem < ns ? rem : ns;
        if (EMIT) {
            uint32_t piece = 0;
            if (lane < cnt) piece = L.exc[i + lane] << L.s16_shift[sel][lane];
            const uint32_t w = wave_or_all(piece) | (sel << 28);
            if (lane == 0) dst[words] = w;
        }
        i += cnt;
        ++words;
    }
    return words;
}

// exception array of the block for width b (FastPFor NewPFor layout): returns nExc, fills L.exc[0 .. 2 nExc)
DS2I_DEV uint32_t optpfor_exceptions(EncLds& L, uint32_t v0, uint32_t v1, uint32_t b) {
    const uint32_t lane = lane_id();
    const bool e0 = (v0 >> b) != 0, e1 = (v1 >> b) != 0;
    const uint64_t m0 = ballot(e0), m1 = ballot(e1);
    const uint32_t n0 = (uint32_t)__builtin_popcountll(m0), n = n0 + (uint32_t)__builtin_popcountll(m1);
    if (!n) return 0;
    const uint64_t lt = (1ull << lane) - 1;
    if (e0) {
        const uint64_t below = m0 & lt;
        const uint32_t idx = (uint32_t)__builtin_popcountll(below);
        const uint32_t delta = below ? lane - (63u - (uint32_t)__builtin_clzll(below)) - 1u : lane;
        L.exc[idx] = delta;
        L.exc[idx + n] = (v0 >> b) - 1u;
    }
    if (e1) {
        const uint64_t below = m1 & lt;
        const uint32_t idx = n0 + (uint32_t)__builtin_popcountll(below);
        uint32_t delta;
        if (below) delta = lane - (63u - (uint32_t)__builtin_clzll(below)) - 1u;
        else if (m0) delta = 64u + lane - (63u - (uint32_t)__builtin_clzll(m0)) - 1u;
        else delta = 64u + lane;
        L.exc[idx] = delta;
        L.exc[idx + n] = (v1 >> b) - 1u;
    }
    wave_sync();
    return n;
}

// findBestB (block_codecs.hpp:156-182). Sizes are in words.
DS2I_DEV uint32_t optpfor_find_best_b(EncLds& L, uint32_t v0, uint32_t v1) {
    const uint32_t orv = wave_or_all(v0 | v1);
    const uint32_t mb = orv ? 32u - (uint32_t)__builtin_clz(orv) : 0u;
    uint32_t i = 0;
    while (mb > 28u + ENC_LOGS[i]) ++i;
    uint32_t best_b = 0, best = 0xFFFFFFFFu;
    for (; i < 17; ++i) {
        const uint32_t b = ENC_LOGS[i];
        if (b > mb) break;
        uint32_t csize;
        if (b == 32) {
            csize = 128;
        } else {
            csize = 4 * b;
            if (csize > best) break; // every later b costs at least its packed words: none can tie or win
            const uint32_t n = optpfor_exceptions(L, v0, v1, b);
            if (n) csize += simple16_words<false>(L, 2 * n, nullptr);
            wave_sync();
        }
        if (csize <= best) { best_b = b; best = csize; }
    }
    return best_b;
}

// bytes of an OptPFor part with width b; with WRITE the dwords are built in L.outw
template <bool WRITE>
DS2I_DEV uint32_t optpfor_part(EncLds& L, uint32_t v0, uint32_t v1, uint32_t b) {
    const uint32_t lane = lane_id();
    if (b == 32) {
        if (WRITE) {
            if (lane == 0) L.outw[0] = 32u << 26;
            L.outw[1 + lane] = v0;
            L.outw[65 + lane] = v1;
            wave_sync();
        }
        return 4 * 129;
    }
    const uint32_t n = optpfor_exceptions(L, v0, v1, b);
    uint32_t ew = 0;
    if (n) ew = simple16_words<WRITE>(L, 2 * n, L.outw + 1);
    if (WRITE) {
        if (lane == 0) L.outw[0] = (b << 26) | (n << 16) | ew;
        uint32_t* pk = L.outw + 1 + ew; // 4 b words: value i occupies bits [i b, (i + 1) b) of the stream
        for (uint32_t w = lane; w < 4 * b; w += 64) pk[w] = 0;
        wave_sync();
        if (b) {
            const uint32_t mask = (uint32_t)((1ull << b) - 1);
            for (int half = 0; half < 2; ++half) {
                const uint32_t idx = lane + 64u * half, x = (half ? v1 : v0) & mask;
                const uint32_t bit = idx * b, sh = bit & 31u;
                atomicOr(&pk[bit >> 5], x << sh);
                if (sh + b > 32) atomicOr(&pk[(bit >> 5) + 1], x >> (32u - sh));
            }
        }
        wave_sync();
    }
    return 4 * (1 + ew + 4 * b);
}

// Interpolative part (blocks of fewer than 128 postings; block_codecs.hpp:105-125, interpolative_coding.hpp:10-77): an
// inherently serial bit stream, written by lane 0 into L.outw (bytes). Returns the byte count.
DS2I_DEV uint32_t interpolative_part(EncLds& L, const uint32_t* vals, uint32_t n, uint32_t sum_of_values) {
    uint32_t bytes = 0;
    if (lane_id() == 0) {
        uint32_t* pre = L.exc; // prefix sums
        pre[0] = vals[0];
        for (uint32_t i = 1; i < n; ++i) pre[i] = pre[i - 1] + vals[i];
        uint8_t* ob = (uint8_t*)L.outw;
        if (sum_of_values == 0xFFFFFFFFu) { // vbyte(sum): 7 bits per byte, the terminator has bit 7 set
            sum_of_values = pre[n - 1];
            uint32_t x = sum_of_values;
            while (x >= 128) { ob[bytes++] = (uint8_t)(x & 127); x >>= 7; }
            ob[bytes++] = (uint8_t)(x | 128);
        }
        // bit writer over 32-bit words (LSB first); the words are assembled at an aligned scratch, then copied behind the vbyte
        uint32_t* wbuf = L.exc + 128;
        uint32_t nw = 0;
        uint64_t size = 0;
        auto write = [&](uint32_t bits, uint32_t len) {
            if (!len) return;
            const uint32_t pos = (uint32_t)(size & 31);
            size += len;
            if (pos == 0) {
                wbuf[nw++] = bits;
            } else {
                wbuf[nw - 1] |= bits << pos;
                if (len > 32 - pos) wbuf[nw++] = bits >> (32 - pos);
            }
        };
        auto write_int = [&](uint32_t val, uint32_t u) { // truncated binary code of val in [0, u)
            const uint32_t b = 31u - (uint32_t)__builtin_clz(u);
            const uint64_t m = (1ull << (b + 1)) - u;
            if (val < m) {
                write(val, b);
            } else {
                val += (uint32_t)m;
                write(val >> 1, b);
                write(val & 1, 1);
            }
        };
        // pre-order walk of write_interpolative(pre, n - 1, 0, sum) with an explicit stack (right child pushed first)
        uint32_t st_off[16], st_cnt[16], st_lo[16], st_hi[16];
        int sp = 0;
        st_off[0] = 0; st_cnt[0] = n - 1; st_lo[0] = 0; st_hi[0] = sum_of_values;
        sp = 1;
        while (sp) {
            --sp;
            uint32_t off = st_off[sp], cnt = st_cnt[sp], lo = st_lo[sp], hi = st_hi[sp];
            while (cnt) { // node, then its left spine; right children wait on the stack
                const uint32_t h = cnt / 2, val = pre[off + h];
                write_int(val - lo, hi - lo + 1);
                if (cnt - h - 1) { st_off[sp] = off + h + 1; st_cnt[sp] = cnt - h - 1; st_lo[sp] = val; st_hi[sp] = hi; ++sp; }
                cnt = h;
                hi = val;
            }
        }
        const uint32_t nb = (uint32_t)((size + 7) / 8);
        const uint8_t* wb = (const uint8_t*)wbuf;
        for (uint32_t i = 0; i < nb; ++i) ob[bytes + i] = wb[i];
        bytes += nb;
    }
    wave_sync();
    return bcast(bytes, 0);
}

DS2I_DEV void copy_out(const EncLds& L, uint8_t* dst, uint32_t bytes) {
    const uint8_t* src = (const uint8_t*)L.outw;
    for (uint32_t i = lane_id(); i < bytes; i += 64) dst[i] = src[i];
}

template <bool WRITE>
__global__ void __launch_bounds__(64) k_encode(EncArgs a) {
    __shared__ EncLds L;
    const uint32_t lane = lane_id();
    enc_tables_init(L);
    for (uint32_t blk = blockIdx.x; blk < a.nblocks; blk += gridDim.x) {
        const uint32_t t = a.blk_list[blk];
        const uint32_t lb = blk - a.list_blk0[t];
        const uint64_t in0 = a.list_in[t];
        const uint32_t n = (uint32_t)(a.list_in[t + 1] - in0);
        const uint64_t k0 = in0 + 128ull * lb;
        const uint32_t sz = n - 128u * lb < 128u ? n - 128u * lb : 128u;
        // gap - 1 and freq - 1, index order (block_posting_list.hpp:31-37)
        const uint32_t d0 = lane < sz ? a.docs[k0 + lane] : 0, d1 = lane + 64 < sz ? a.docs[k0 + 64 + lane] : 0;
        const uint32_t prev_last = lb ? a.docs[k0 - 1] : 0xFFFFFFFFu;
        L.v[1][lane] = d0;
        L.v[1][lane + 64] = d1;
        wave_sync();
        const uint32_t p0 = lane ? L.v[1][lane - 1] : prev_last, p1 = L.v[1][lane + 63];
        const uint32_t g0 = lane < sz ? d0 - p0 - 1u : 0u, g1 = lane + 64 < sz ? d1 - p1 - 1u : 0u;
        const uint32_t f0 = lane < sz ? a.freqs[k0 + lane] - 1u : 0u, f1 = lane + 64 < sz ? a.freqs[k0 + 64 + lane] - 1u : 0u;
        const uint32_t last_doc = uniform(L.v[1][sz - 1]);
        wave_sync();
        L.v[0][lane] = g0;
        L.v[0][lane + 64] = g1;
        L.v[1][lane] = f0;
        L.v[1][lane + 64] = f1;
        wave_sync();
        const uint32_t block_base = lb ? prev_last + 1u : 0u;
        uint8_t* dst = nullptr;
        if (WRITE) dst = a.out + a.blk_out[blk];
        for (int part = 0; part < 2; ++part) {
            const uint32_t v0 = part ? f0 : g0, v1 = part ? f1 : g1;
            uint32_t bytes;
            if (sz == 128) {
                uint32_t b;
                if (WRITE) b = a.bsel[2ull * blk + part];
                else b = optpfor_find_best_b(L, v0, v1);
                bytes = optpfor_part<WRITE>(L, v0, v1, b);
                if (!WRITE && lane == 0) a.bsel[2ull * blk + part] = (uint8_t)b;
            } else {
                bytes = interpolative_part(L, L.v[part], sz, part ? 0xFFFFFFFFu : last_doc - block_base - (sz - 1));
            }
            if (WRITE) {
                copy_out(L, dst, bytes);
                dst += bytes;
                wave_sync();
            } else if (lane == 0) {
                a.psize[2ull * blk + part] = bytes;
            }
        }
        if (!WRITE) {
            if (lane == 0) a.bmax[blk] = last_doc;
        } else if (lane == 0) {
            // list header: vbyte(n) | block_max[nb] | block_endpoint[nb - 1]
            const uint32_t nb = (n + 127u) >> 7;
            uint8_t* lp = a.out + a.list_out[t];
            uint32_t vl = 0;
            {
                uint32_t x = n;
                while (x >= 128) { if (lb == 0) lp[vl] = (uint8_t)(x & 127); ++vl; x >>= 7; }
                if (lb == 0) lp[vl] = (uint8_t)(x | 128);
                ++vl;
            }
            uint8_t* maxs = lp + vl;
            uint8_t* eps = maxs + 4ull * nb;
            const uint8_t* blocks = eps + 4ull * (nb - 1);
            __builtin_memcpy(maxs + 4ull * lb, &last_doc, 4);
            if (lb + 1 < nb) { // offset of the NEXT block inside the blocks area
                const uint32_t ep = (uint32_t)((a.out + a.blk_out[blk + 1]) - blocks);
                __builtin_memcpy(eps + 4ull * lb, &ep, 4);
            }
        }
        wave_sync();
    }
}

} // namespace

extern "C" {
size_t ds2i_sizeof_enc_args() { return sizeof(EncArgs); }
hipError_t ds2i_launch_encode(int write, const void* args, unsigned grid, hipStream_t s) {
    const EncArgs& a = *(const EncArgs*)args;
    if (write) hipLaunchKernelGGL(k_encode<true>, dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_encode<false>, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}
}
