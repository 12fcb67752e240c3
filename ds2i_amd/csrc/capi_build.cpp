// Host-side (CPU) implementation of include/ds2i_build.h. No GPU work here.
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/ds2i_build.h"
#include "capi_blob.hpp"
#include "capi_error.hpp"
#include "host_encode.hpp"
#include "host_index.hpp"
#include "host_pef.hpp"
#include "host_hybrid.hpp"
#include "host_synth.hpp"

using namespace ds2i_host;

struct ds2i_builder {
    std::unique_ptr<block_index_builder> b;   // kinds 0..4
    std::unique_ptr<opt_index_builder> opt;   // kinds 5..8 (DS2I_OPT / EF / SINGLE / UNIFORM)
};
struct ds2i_hybrid {
    std::unique_ptr<hybrid_index_builder> b;
};
struct ds2i_wand_builder {
    std::vector<float> norm_lens, max_w;
};

namespace {
synth_params to_params(const ds2i_synth_params* p) {
    synth_params s;
    s.seed = p->seed; s.num_docs = p->num_docs; s.num_terms = p->num_terms; s.zipf_exp = p->zipf_exp;
    s.top_df_frac = p->top_df_frac; s.min_len = p->min_len; s.clustered_every = p->clustered_every;
    s.topics = p->topics; s.topic_boost = p->topic_boost;
    return s;
}
} // namespace

#define DS2I_TRY try {
#define DS2I_CATCH                                                                       \
    } catch (std::bad_alloc const&) { return ds2i_set_error(-7, "out of memory"); }     \
    catch (std::invalid_argument const& e) { return ds2i_set_error(-1, e.what()); }      \
    catch (std::exception const& e) { return ds2i_set_error(-2, e.what()); }

extern "C" {

const uint8_t* ds2i_blob_data(const ds2i_blob* b) { return b ? b->data.data() : nullptr; }
size_t ds2i_blob_size(const ds2i_blob* b) { return b ? b->data.size() : 0; }
void ds2i_blob_free(ds2i_blob* b) { delete b; }

int ds2i_builder_create(int codec, uint64_t num_docs, ds2i_builder** out) {
    if (!out || codec < 0 || codec > LAYOUT_UNIFORM) return ds2i_set_error(-1, "ds2i_builder_create: bad argument");
    DS2I_TRY
    auto* h = new ds2i_builder;
    if (is_freq_layout(codec)) h->opt.reset(new opt_index_builder(num_docs, global_parameters(), codec));
    else h->b.reset(new block_index_builder(codec, num_docs));
    *out = h;
    return 0;
    DS2I_CATCH
}
int ds2i_builder_add_posting_list(ds2i_builder* b, uint64_t n, const uint32_t* docs, const uint32_t* freqs) {
    if (!b || !docs || !freqs) return ds2i_set_error(-1, "ds2i_builder_add_posting_list: null argument");
    DS2I_TRY
    if (b->opt) b->opt->add_posting_list(n, docs, freqs);
    else b->b->add_posting_list(n, docs, freqs);
    return 0;
    DS2I_CATCH
}
int ds2i_builder_freeze(ds2i_builder* b, ds2i_blob** image) {
    if (!b || !image) return ds2i_set_error(-1, "ds2i_builder_freeze: null argument");
    DS2I_TRY
    auto* blob = new ds2i_blob;
    if (b->opt) b->opt->freeze(blob->data);
    else b->b->freeze(blob->data);
    *image = blob;
    return 0;
    DS2I_CATCH
}
void ds2i_builder_free(ds2i_builder* b) { delete b; }

int ds2i_wand_create(const uint32_t* doc_sizes, uint64_t num_docs, ds2i_wand_builder** out) {
    if (!doc_sizes || !out || !num_docs) return ds2i_set_error(-1, "ds2i_wand_create: bad argument");
    DS2I_TRY
    auto* w = new ds2i_wand_builder;
    compute_norm_lens(doc_sizes, num_docs, w->norm_lens);
    *out = w;
    return 0;
    DS2I_CATCH
}
int ds2i_wand_add_list(ds2i_wand_builder* w, uint64_t n, const uint32_t* docs, const uint32_t* freqs) {
    if (!w || !docs || !freqs) return ds2i_set_error(-1, "ds2i_wand_add_list: null argument");
    for (uint64_t i = 0; i < n; ++i)
        if (docs[i] >= w->norm_lens.size()) return ds2i_set_error(-1, "ds2i_wand_add_list: doc id out of range");
    w->max_w.push_back(list_max_weight(w->norm_lens.data(), n, docs, freqs));
    return 0;
}
int ds2i_wand_freeze(ds2i_wand_builder* w, ds2i_blob** image) {
    if (!w || !image) return ds2i_set_error(-1, "ds2i_wand_freeze: null argument");
    DS2I_TRY
    auto* blob = new ds2i_blob;
    wand_freeze(w->norm_lens, w->max_w, blob->data);
    *image = blob;
    return 0;
    DS2I_CATCH
}
void ds2i_wand_free(ds2i_wand_builder* w) { delete w; }

int ds2i_encode_block(int codec, const uint32_t* values, uint32_t sum, uint32_t n, ds2i_blob** out) {
    if (!values || !out || n == 0 || n > BLOCK) return ds2i_set_error(-1, "ds2i_encode_block: bad argument");
    DS2I_TRY
    auto* blob = new ds2i_blob;
    block_encode(codec, values, sum, n, blob->data);
    *out = blob;
    return 0;
    DS2I_CATCH
}
// the host half of the BM25 scorer exactly as the query path uses it (query weights in plan_batch, max_term_weight in
// the wand builder); element-wise so that tests can hold it against the reference's bm25.hpp
int ds2i_bm25_query_term_weight(const uint64_t* qtf, const uint64_t* df, uint64_t num_docs, uint64_t n, float* out) {
    if (!qtf || !df || !out) return ds2i_set_error(-1, "ds2i_bm25_query_term_weight: null argument");
    for (uint64_t i = 0; i < n; ++i) out[i] = ds2i_host::bm25::query_term_weight(qtf[i], df[i], num_docs);
    return 0;
}
int ds2i_bm25_doc_term_weight(const uint64_t* freq, const float* norm_len, uint64_t n, float* out) {
    if (!freq || !norm_len || !out) return ds2i_set_error(-1, "ds2i_bm25_doc_term_weight: null argument");
    for (uint64_t i = 0; i < n; ++i) out[i] = ds2i_host::bm25::doc_term_weight(freq[i], norm_len[i]);
    return 0;
}

int ds2i_write_sequence(int seq_kind, const uint64_t* values, uint64_t n, uint64_t universe, const uint8_t params[5],
                        ds2i_blob** bits, uint64_t* nbits) {
    if (!values || !bits || !nbits || !n) return ds2i_set_error(-1, "ds2i_write_sequence: bad argument");
    DS2I_TRY
    ds2i_host::global_parameters gp;
    if (params) {
        gp.ef_log_sampling0 = params[0];
        gp.ef_log_sampling1 = params[1];
        gp.rb_log_rank1_sampling = params[2];
        gp.rb_log_sampling1 = params[3];
        gp.log_partition_size = params[4];
    }
    ds2i_host::bitvec_builder bvb;
    switch (seq_kind) {
    case DS2I_SEQ_ELIAS_FANO: ds2i_host::ef_write(bvb, values, universe, n, gp); break;
    case DS2I_SEQ_RANKED_BITVECTOR: ds2i_host::rb_write(bvb, values, universe, n, gp); break;
    case DS2I_SEQ_INDEXED: ds2i_host::seq_write<false>(bvb, values, universe, n, gp); break;
    case DS2I_SEQ_STRICT: ds2i_host::seq_write<true>(bvb, values, universe, n, gp); break;
    case DS2I_SEQ_PARTITIONED_INDEXED: ds2i_host::partitioned_write<false>(bvb, values, universe, n, gp); break;
    case DS2I_SEQ_PARTITIONED_STRICT: ds2i_host::partitioned_write<true>(bvb, values, universe, n, gp); break;
    case DS2I_SEQ_UNIFORM_INDEXED: ds2i_host::uniform_write<false>(bvb, values, universe, n, gp); break;
    case DS2I_SEQ_UNIFORM_STRICT: ds2i_host::uniform_write<true>(bvb, values, universe, n, gp); break;
    default: return ds2i_set_error(-1, "ds2i_write_sequence: unknown sequence kind");
    }
    auto* blob = new ds2i_blob;
    const uint8_t* w = (const uint8_t*)bvb.words().data();
    blob->data.assign(w, w + 8 * bvb.words().size());
    *nbits = bvb.size();
    *bits = blob;
    return 0;
    DS2I_CATCH
}

int ds2i_encode_vbyte(uint32_t value, ds2i_blob** out) {
    if (!out) return ds2i_set_error(-1, "ds2i_encode_vbyte: null argument");
    auto* blob = new ds2i_blob;
    vbyte_encode(value, blob->data);
    *out = blob;
    return 0;
}
int ds2i_encode_posting_list(int codec, uint32_t n, const uint32_t* docs, const uint32_t* freqs, ds2i_blob** out) {
    if (!docs || !freqs || !out || !n) return ds2i_set_error(-1, "ds2i_encode_posting_list: bad argument");
    DS2I_TRY
    auto* blob = new ds2i_blob;
    write_posting_list(codec, blob->data, n, docs, freqs);
    *out = blob;
    return 0;
    DS2I_CATCH
}

int ds2i_opt_list_directory(const void* opt_image, size_t bytes, uint32_t term, ds2i_blob** cmax, ds2i_blob** chunks,
                            uint64_t info[5]) {
    return ds2i_freq_list_directory(LAYOUT_OPT, opt_image, bytes, term, cmax, chunks, info);
}
int ds2i_freq_list_directory(int kind, const void* opt_image, size_t bytes, uint32_t term, ds2i_blob** cmax,
                             ds2i_blob** chunks, uint64_t info[5]) {
    if (!opt_image || !cmax || !chunks || !info || !is_freq_layout(kind))
        return ds2i_set_error(-1, "ds2i_freq_list_directory: bad argument");
    DS2I_TRY
    opt_index_view v;
    v.layout = kind;
    v.parse(opt_image, bytes);
    if (term >= v.size) return ds2i_set_error(-3, "term id out of range");
    pef_list_dir dir;
    v.build_dir(term, dir);
    auto* bc = new ds2i_blob;
    auto* bk = new ds2i_blob;
    const uint8_t* p = (const uint8_t*)dir.cmax.data();
    bc->data.assign(p, p + 4 * dir.cmax.size());
    p = (const uint8_t*)dir.chunks.data();
    bk->data.assign(p, p + sizeof(pef_chunk) * dir.chunks.size());
    *cmax = bc;
    *chunks = bk;
    info[0] = dir.n;
    info[1] = dir.docs_bit0;
    info[2] = dir.freqs_bit0;
    info[3] = (uint64_t)(v.docs_bits.bytes - (const uint8_t*)opt_image);
    info[4] = (uint64_t)(v.freqs_bits.bytes - (const uint8_t*)opt_image);
    return 0;
    DS2I_CATCH
}

uint64_t ds2i_synth_list_upper_bound(const ds2i_synth_params* p, uint32_t term) {
    (void)term;
    return p ? p->num_docs : 0;
}
int ds2i_synth_list(const ds2i_synth_params* p, uint32_t term, uint32_t* docs, uint32_t* freqs, uint64_t capacity,
                    uint64_t* n) {
    if (!p || !docs || !freqs || !n) return ds2i_set_error(-1, "ds2i_synth_list: null argument");
    DS2I_TRY
    std::vector<uint32_t> d, f;
    uint64_t len = synth_list(to_params(p), term, d, f);
    *n = len;
    if (len > capacity) return ds2i_set_error(-1, "ds2i_synth_list: capacity too small");
    std::memcpy(docs, d.data(), 4 * len);
    std::memcpy(freqs, f.data(), 4 * len);
    return 0;
    DS2I_CATCH
}
int ds2i_synth_doc_sizes(const ds2i_synth_params* p, uint32_t* sizes) {
    if (!p || !sizes) return ds2i_set_error(-1, "ds2i_synth_doc_sizes: null argument");
    DS2I_TRY
    std::vector<uint32_t> s;
    synth_doc_sizes(to_params(p), s);
    std::memcpy(sizes, s.data(), 4 * s.size());
    return 0;
    DS2I_CATCH
}
int ds2i_synth_queries(uint64_t seed, uint32_t num_terms, uint32_t nq, uint32_t* terms, uint32_t* offsets) {
    if (!terms || !offsets || !num_terms) return ds2i_set_error(-1, "ds2i_synth_queries: bad argument");
    DS2I_TRY
    std::vector<uint32_t> t, o;
    synth_queries(seed, num_terms, nq, t, o);
    std::memcpy(terms, t.data(), 4 * t.size());
    std::memcpy(offsets, o.data(), 4 * o.size());
    return 0;
    DS2I_CATCH
}

int ds2i_synth_queries_topical(const ds2i_synth_params* pp, uint64_t seed, uint32_t nq, uint32_t same_topic_pct, uint32_t* terms, uint32_t* offsets) {
    if (!pp || !terms || !offsets || !pp->num_terms) return ds2i_set_error(-1, "ds2i_synth_queries_topical: bad argument");
    DS2I_TRY
    std::vector<uint32_t> t, o;
    synth_queries_topical(to_params(pp), seed, nq, same_topic_pct, t, o);
    std::memcpy(terms, t.data(), 4 * t.size());
    std::memcpy(offsets, o.data(), 4 * o.size());
    return 0;
    DS2I_CATCH
}

int ds2i_synth_build(const ds2i_synth_params* pp, int codec, int threads, ds2i_blob** index_image,
                     ds2i_blob** wand_image, uint64_t* total_postings) {
    if (!pp || !index_image || codec < 0 || codec > LAYOUT_UNIFORM) return ds2i_set_error(-1, "ds2i_synth_build: bad argument");
    const bool freq_layout = is_freq_layout(codec);
    DS2I_TRY
    const synth_params p = to_params(pp);
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    std::vector<uint32_t> sizes;
    synth_doc_sizes(p, sizes);
    std::vector<float> norm_lens;
    compute_norm_lens(sizes.data(), p.num_docs, norm_lens);
    sizes.clear();
    sizes.shrink_to_fit();
    const uint32_t V = p.num_terms;
    std::vector<bytes_t> enc(V);
    std::vector<bitvec_builder> enc_docs(freq_layout ? V : 0), enc_freqs(freq_layout ? V : 0);
    std::vector<float> max_w(V);
    std::atomic<uint32_t> next(0);
    std::atomic<uint64_t> postings(0);
    std::string err;
    std::mutex err_mu;
    auto worker = [&]() {
        std::vector<uint32_t> d, f;
        try {
            for (;;) {
                uint32_t t = next.fetch_add(1);
                if (t >= V) break;
                uint64_t n = synth_list(p, t, d, f);
                if (freq_layout) opt_index_builder::encode_list(p.num_docs, global_parameters(), n, d.data(), f.data(), enc_docs[t], enc_freqs[t], codec);
                else write_posting_list(codec, enc[t], (uint32_t)n, d.data(), f.data());
                max_w[t] = list_max_weight(norm_lens.data(), n, d.data(), f.data());
                postings += n;
            }
        } catch (std::exception const& e) {
            std::lock_guard<std::mutex> g(err_mu);
            err = e.what();
        }
    };
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
    for (auto& th : pool) th.join();
    if (!err.empty()) return ds2i_set_error(-2, err.c_str());
    auto* ib = new ds2i_blob;
    if (freq_layout) {
        opt_index_builder builder(p.num_docs, global_parameters(), codec);
        for (uint32_t t = 0; t < V; ++t) {
            builder.add_encoded(enc_docs[t], enc_freqs[t]);
            enc_docs[t] = bitvec_builder();
            enc_freqs[t] = bitvec_builder();
        }
        builder.freeze(ib->data);
    } else {
        block_index_builder builder(codec, p.num_docs);
        for (uint32_t t = 0; t < V; ++t) {
            builder.add_encoded_list(enc[t].data(), enc[t].size());
            bytes_t().swap(enc[t]);
        }
        builder.freeze(ib->data);
    }
    *index_image = ib;
    if (wand_image) {
        auto* wb = new ds2i_blob;
        wand_freeze(norm_lens, max_w, wb->data);
        *wand_image = wb;
    }
    if (total_postings) *total_postings = postings;
    return 0;
    DS2I_CATCH
}


// ---------------------------------------------------------------- block_mixed optimiser (host_hybrid.hpp)
void ds2i_hybrid_default_model(ds2i_hybrid_model* m) {
    if (!m) return;
    hybrid_model d;
    m->pfor_base = d.pfor_base; m->pfor_exc = d.pfor_exc; m->pfor_exc_many = d.pfor_exc_many;
    m->varint = d.varint; m->interp_base = d.interp_base; m->interp_node = d.interp_node;
}
int ds2i_hybrid_create(uint64_t num_docs, const ds2i_hybrid_model* model, ds2i_hybrid** out) {
    if (!out) return ds2i_set_error(-1, "ds2i_hybrid_create: null argument");
    DS2I_TRY
    hybrid_model m;
    if (model) {
        m.pfor_base = model->pfor_base; m.pfor_exc = model->pfor_exc; m.pfor_exc_many = model->pfor_exc_many;
        m.varint = model->varint; m.interp_base = model->interp_base; m.interp_node = model->interp_node;
    }
    auto* h = new ds2i_hybrid;
    h->b.reset(new hybrid_index_builder(num_docs, m));
    *out = h;
    return 0;
    DS2I_CATCH
}
int ds2i_hybrid_add_posting_list(ds2i_hybrid* h, uint64_t n, const uint32_t* docs, const uint32_t* freqs, const uint32_t* access) {
    if (!h || !docs || !freqs) return ds2i_set_error(-1, "ds2i_hybrid_add_posting_list: null argument");
    DS2I_TRY
    h->b->add_posting_list(n, docs, freqs, access);
    return 0;
    DS2I_CATCH
}
int ds2i_hybrid_analyse(ds2i_hybrid* h, int threads, uint64_t* min_space, uint64_t* max_space) {
    if (!h) return ds2i_set_error(-1, "ds2i_hybrid_analyse: null argument");
    DS2I_TRY
    if (!h->b->analysed()) h->b->analyse(threads);
    if (min_space) *min_space = h->b->min_space();
    if (max_space) *max_space = h->b->max_space();
    return 0;
    DS2I_CATCH
}
int ds2i_hybrid_freeze(ds2i_hybrid* h, uint64_t budget_bytes, int threads, ds2i_blob** image, double* rate, uint64_t* space,
                       double* model_time, uint64_t type_counts[6]) {
    if (!h || !image) return ds2i_set_error(-1, "ds2i_hybrid_freeze: null argument");
    DS2I_TRY
    if (!h->b->analysed()) h->b->analyse(threads);
    if (budget_bytes < h->b->min_space()) return ds2i_set_error(-1, "budget below the smallest possible index");
    const double r = h->b->solve(budget_bytes);
    uint64_t s = 0;
    double t = 0;
    h->b->evaluate(r, s, t);
    auto* blob = new ds2i_blob;
    h->b->freeze(r, threads, blob->data, type_counts);
    *image = blob;
    if (rate) *rate = r;
    if (space) *space = s;
    if (model_time) *model_time = t;
    return 0;
    DS2I_CATCH
}
void ds2i_hybrid_free(ds2i_hybrid* h) { delete h; }

// The synthetic collection through the optimiser: lists are regenerated (pure functions of seed and term) for the
// analysis pass and again for the encoding pass, so a GOV2- / ClueWeb-scale collection never sits in memory raw.
// budget = min_space + budget_frac * (max_space - min_space).
int ds2i_synth_build_hybrid(const ds2i_synth_params* pp, int threads, const ds2i_hybrid_model* model, const uint32_t* access,
                            double budget_frac, ds2i_blob** index_image, ds2i_blob** wand_image, uint64_t* total_postings,
                            uint64_t type_counts[6]) {
    if (!pp || !index_image || !(budget_frac >= 0.0 && budget_frac <= 1.0))
        return ds2i_set_error(-1, "ds2i_synth_build_hybrid: bad argument");
    DS2I_TRY
    const synth_params p = to_params(pp);
    if (threads <= 0) threads = (int)std::max(1u, std::thread::hardware_concurrency());
    hybrid_model m;
    if (model) {
        m.pfor_base = model->pfor_base; m.pfor_exc = model->pfor_exc; m.pfor_exc_many = model->pfor_exc_many;
        m.varint = model->varint; m.interp_base = model->interp_base; m.interp_node = model->interp_node;
    }
    const uint32_t V = p.num_terms;
    std::vector<uint32_t> sizes;
    synth_doc_sizes(p, sizes);
    std::vector<float> norm_lens;
    compute_norm_lens(sizes.data(), p.num_docs, norm_lens);
    std::vector<uint32_t>().swap(sizes);
    // pass 0: list lengths (block numbering of `access`), max term weights
    std::vector<float> max_w(V, 0.f);
    std::vector<uint64_t> len(V, 0);
    {
        std::atomic<uint32_t> next(0);
        auto worker = [&]() {
            std::vector<uint32_t> d, f;
            for (;;) {
                const uint32_t t = next.fetch_add(1);
                if (t >= V) break;
                len[t] = synth_list(p, t, d, f);
                max_w[t] = list_max_weight(norm_lens.data(), len[t], d.data(), f.data());
            }
        };
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    }
    hybrid_index_builder hb(p.num_docs, m);
    uint64_t base = 0, postings = 0;
    for (uint32_t t = 0; t < V; ++t) {
        const uint64_t blocks = ceil_div(len[t], (uint64_t)BLOCK);
        hb.add_virtual_list(access ? access + 2 * base : nullptr, blocks);
        base += blocks;
        postings += len[t];
    }
    hb.set_provider([&](size_t t, std::vector<uint32_t>& d, std::vector<uint32_t>& f) {
        const uint64_t n = synth_list(p, (uint32_t)t, d, f);
        d.resize(n);
        f.resize(n);
    });
    hb.analyse(threads);
    const uint64_t lo = hb.min_space(), hi = hb.max_space();
    const uint64_t budget = lo + (uint64_t)(budget_frac * double(hi - lo));
    const double rate = hb.solve(budget);
    auto* ib = new ds2i_blob;
    hb.freeze(rate, threads, ib->data, type_counts);
    *index_image = ib;
    if (wand_image) {
        auto* wb = new ds2i_blob;
        wand_freeze(norm_lens, max_w, wb->data);
        *wand_image = wb;
    }
    if (total_postings) *total_postings = postings;
    return 0;
    DS2I_CATCH
}

} // extern "C"
