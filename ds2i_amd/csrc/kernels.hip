// HIP class kernels (gfx950 / CDNA4, wave64) of the ds2i batched query path: everything the stream kernels (ranked_stream.hip,
// union_stream.hip, freq_stream.hip) do not answer, the reference-order traversals, and the upload-time passes.
// One wavefront per WORK UNIT (a piece of a query: a block range of its shortest list, or a doc-id range), one unit per single-wave
// workgroup; every memory operation is wave-cooperative, control flow is wave-uniform. No MFMA (integer work).
// The kernels live in one include file per operator family; this file holds the launchers (called from capi*.cpp):
//   kernels_common.inc        per-wave LDS layout, enumerator construction, top-k stores
//   kernels_conjunctive.inc   k_conjunctive  and_query / ranked_and_query (queries.hpp:35-86, 322-401), block-synchronous; k_merge
//   kernels_daat.inc          k_daat / k_daat_long  every operator in the reference's one-document-per-step order
//                             (DS2I_OP_REFERENCE_ORDER; > 16 terms; k > 64)
//   kernels_disjunctive.inc   k_disjunctive, k_union_topk  wand / maxscore / ranked_or (queries.hpp:200-319, 404-476, 478-591) without
//                             side slots / range tables; k_union  or / or_freq (queries.hpp:88-131) as a stream
//   kernels_upload.inc        k_decode_list[_side], k_block_max_weights, k_build_side_tables, k_list_top_bmw, self-tests
// Compiled once per list-count class (-DDS2I_TU_TMAX) and once for everything else: ds2i_amd/build.py.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_enum.hpp"
#include "device_score.hpp"

using namespace ds2i_dev;

namespace {
#include "kernels_common.inc"
#include "kernels_conjunctive.inc"
#include "kernels_daat.inc"
#include "kernels_disjunctive.inc"
#include "kernels_upload.inc"

} // namespace

// ------------------------------------------------------------------ launchers (called from capi.cpp)
// the block_optpfor specialisations decode through the upload-time side tables; an index uploaded without them runs the
// runtime-codec instantiations
static inline bool optpfor_side(const BatchArgs& a) { return a.codec == CODEC_OPTPFOR && a.xslots != nullptr && a.tails != nullptr; }

namespace ds2i_launch {

struct Batch {
    BatchArgs a;
};

// The file is compiled once per list-count class (-DDS2I_TU_TMAX=2|4|8|16: only launch_t<TMAX> and the kernels it
// instantiates) and once without the macro (everything else): five translation units that build.py compiles in
// parallel -- the kernel templates are by far the slowest part of the build.
// DS2I_TU_TMAX == 0: the translation unit of the "long" class (k_daat_long, every operator in reference order)
#if defined(DS2I_TU_TMAX) && DS2I_TU_TMAX == 0
hipError_t launch_long(int op, const BatchArgs& a, unsigned grid, hipStream_t s) {
    dim3 g(grid), b(64);
    if (a.k > 64) { // top-k beyond one score per lane: 16 scores per lane (k <= 1024)
        typedef TopKBig<16> BIG;
        switch (op & 0xFF) {
        case OP_RANKED_AND: hipLaunchKernelGGL((k_daat_long<OP_RANKED_AND, BIG>), g, b, 0, s, a); break;
        case OP_WAND: hipLaunchKernelGGL((k_daat_long<OP_WAND, BIG>), g, b, 0, s, a); break;
        case OP_MAXSCORE: hipLaunchKernelGGL((k_daat_long<OP_MAXSCORE, BIG>), g, b, 0, s, a); break;
        case OP_RANKED_OR: hipLaunchKernelGGL((k_daat_long<OP_RANKED_OR, BIG>), g, b, 0, s, a); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (op & 0xFF) {
    case OP_AND: hipLaunchKernelGGL((k_daat_long<OP_AND>), g, b, 0, s, a); break;
    case OP_AND_FREQ: hipLaunchKernelGGL((k_daat_long<OP_AND_FREQ>), g, b, 0, s, a); break;
    case OP_OR: hipLaunchKernelGGL((k_daat_long<OP_OR>), g, b, 0, s, a); break;
    case OP_OR_FREQ: hipLaunchKernelGGL((k_daat_long<OP_OR_FREQ>), g, b, 0, s, a); break;
    case OP_RANKED_AND: hipLaunchKernelGGL((k_daat_long<OP_RANKED_AND>), g, b, 0, s, a); break;
    case OP_WAND: hipLaunchKernelGGL((k_daat_long<OP_WAND>), g, b, 0, s, a); break;
    case OP_MAXSCORE: hipLaunchKernelGGL((k_daat_long<OP_MAXSCORE>), g, b, 0, s, a); break;
    case OP_RANKED_OR: hipLaunchKernelGGL((k_daat_long<OP_RANKED_OR>), g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
#else
hipError_t launch_long(int op, const BatchArgs& a, unsigned grid, hipStream_t s);
#endif

template <int TMAX>
hipError_t launch_t(int op, const BatchArgs& a, unsigned grid, hipStream_t s)
#if defined(DS2I_TU_TMAX) && DS2I_TU_TMAX > 0
{
    dim3 g(grid), b(64);
    const size_t dyn = 1024u * (size_t)a.dyn_lists; // union kernels: docs + freqs of dyn_lists list slots
    switch (op) {
    // the conjunctive kernels are specialised for block_optpfor (the benchmark codec), the freq_index family and
    // block_mixed (configs[4]; its three block types stay a run-time switch, QMX drops out); block_varint /
    // block_interpolative / block_qmx go through the runtime-dispatch instantiation (CODEC_T = -1)
    case OP_AND:
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_PEF, false>), g, b, 0, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_PEF>), g, b, 0, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, CODEC_MIXED>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_conjunctive<false, false, TMAX, -1>), g, b, 0, s, a);
        break;
    case OP_AND_FREQ:
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF && !a.stats) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_PEF, false>), g, b, 0, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_PEF>), g, b, 0, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, CODEC_MIXED>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_conjunctive<false, true, TMAX, -1>), g, b, 0, s, a);
        break;
    case OP_RANKED_AND:
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF && !a.stats) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_PEF, false>), g, b, 0, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_PEF>), g, b, 0, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, CODEC_MIXED>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_conjunctive<true, true, TMAX, -1>), g, b, 0, s, a);
        break;
    // the ranked disjunctive operators get the same two codec specialisations (BASELINE configs[3] runs them on
    // block_optpfor); or / or_freq and the reference-order conjunctions stay on the runtime-dispatch instantiation
    // (or / or_freq run k_union for every list count -- ds2i_launch_batch below; the windowed MODE 1 / 2 instantiations of k_disjunctive
    // that answered them until round 3 are no longer built)
    case 0x100 | OP_OR: hipLaunchKernelGGL((k_daat<OP_OR, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_OR_FREQ: hipLaunchKernelGGL((k_daat<OP_OR_FREQ, TMAX>), g, b, 0, s, a); break;
    // wand / maxscore / ranked_or: the block-synchronous disjunctive kernel (identical results by definition)
    case OP_WAND:
    case OP_MAXSCORE:
    case OP_RANKED_OR:
        if (a.vq_info) { // the streaming form (units = (query, driving list, block range)); needs the range tables
            if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_union_topk<TMAX, CODEC_OPTPFOR, false>), g, b, 0, s, a);
            else if (optpfor_side(a)) hipLaunchKernelGGL((k_union_topk<TMAX, CODEC_OPTPFOR>), g, b, 0, s, a);
            else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_union_topk<TMAX, CODEC_PEF>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union_topk<TMAX, -1>), g, b, 0, s, a);
            break;
        }
        if (optpfor_side(a) && !a.stats) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_OPTPFOR, false>), g, b, dyn, s, a);
        else if (optpfor_side(a)) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_OPTPFOR>), g, b, dyn, s, a);
        else if (a.codec == CODEC_PEF) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_PEF>), g, b, dyn, s, a);
        else if (a.codec == CODEC_MIXED) hipLaunchKernelGGL((k_disjunctive<TMAX, CODEC_MIXED>), g, b, dyn, s, a);
        else hipLaunchKernelGGL((k_disjunctive<TMAX, -1>), g, b, dyn, s, a);
        break;
    // reference-order (one document per step) traversals of the same operators: op | OP_REFERENCE_ORDER
    case 0x100 | OP_WAND: hipLaunchKernelGGL((k_daat<OP_WAND, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_MAXSCORE: hipLaunchKernelGGL((k_daat<OP_MAXSCORE, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_RANKED_OR: hipLaunchKernelGGL((k_daat<OP_RANKED_OR, TMAX>), g, b, 0, s, a); break;
    // reference-order (one candidate per step) conjunctive traversal: op | OP_REFERENCE_ORDER
    case 0x100 | OP_AND: hipLaunchKernelGGL((k_daat<OP_AND, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_AND_FREQ: hipLaunchKernelGGL((k_daat<OP_AND_FREQ, TMAX>), g, b, 0, s, a); break;
    case 0x100 | OP_RANKED_AND: hipLaunchKernelGGL((k_daat<OP_RANKED_AND, TMAX>), g, b, 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template hipError_t launch_t<DS2I_TU_TMAX>(int, const BatchArgs&, unsigned, hipStream_t);
#else
;
extern template hipError_t launch_t<2>(int, const BatchArgs&, unsigned, hipStream_t);
extern template hipError_t launch_t<4>(int, const BatchArgs&, unsigned, hipStream_t);
extern template hipError_t launch_t<8>(int, const BatchArgs&, unsigned, hipStream_t);
extern template hipError_t launch_t<16>(int, const BatchArgs&, unsigned, hipStream_t);
#endif

} // namespace ds2i_launch

#if !defined(DS2I_TU_TMAX)
extern "C" {

// tmax_class: 0 -> TMAX 2, 1 -> TMAX 4, 2 -> TMAX 8, 3 -> TMAX 16 (LDS footprint per wave grows with TMAX),
// 4 -> more than 16 terms (state in global scratch)
hipError_t ds2i_launch_batch(int op, int tmax_class, const void* args, unsigned grid, hipStream_t s) {
    const BatchArgs& a = *(const BatchArgs*)args;
    if ((op == OP_OR || op == OP_OR_FREQ) && a.dyn_lists == 0xFFFFFFFFu) { // or_query as a stream: one kernel for every list count
        const dim3 g(grid), b(64);
        const bool f = op == OP_OR_FREQ;
        if (optpfor_side(a) && !a.stats) {
            if (f) hipLaunchKernelGGL((k_union<true, CODEC_OPTPFOR, false>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union<false, CODEC_OPTPFOR, false>), g, b, 0, s, a);
        } else if (optpfor_side(a)) {
            if (f) hipLaunchKernelGGL((k_union<true, CODEC_OPTPFOR>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union<false, CODEC_OPTPFOR>), g, b, 0, s, a);
        } else {
            if (f) hipLaunchKernelGGL((k_union<true, -1>), g, b, 0, s, a);
            else hipLaunchKernelGGL((k_union<false, -1>), g, b, 0, s, a);
        }
        return hipGetLastError();
    }
    switch (tmax_class) {
    case 0: return ds2i_launch::launch_t<2>(op, a, grid, s);
    case 1: return ds2i_launch::launch_t<4>(op, a, grid, s);
    case 2: return ds2i_launch::launch_t<8>(op, a, grid, s);
    case 3: return ds2i_launch::launch_t<16>(op, a, grid, s);
    default: return ds2i_launch::launch_long(op, a, grid, s);
    }
}

uint32_t ds2i_meta_words(void) { return (uint32_t)M_WORDS; }

hipError_t ds2i_launch_block_max_weights(const void* args, unsigned grid, hipStream_t s) {
    const BmwArgs& a = *(const BmwArgs*)args;
    hipLaunchKernelGGL(k_block_max_weights, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_build_side_tables(const void* args, unsigned grid, hipStream_t s) {
    const SideArgs& a = *(const SideArgs*)args;
    hipLaunchKernelGGL(k_build_side_tables, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t ds2i_launch_list_top_bmw(const float* bmw, const void* lists, uint32_t nlists, float* out, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL(k_list_top_bmw, dim3(grid), dim3(64), 0, s, bmw, (const QTerm*)lists, nlists, out);
    return hipGetLastError();
}

hipError_t ds2i_launch_merge(const void* args, unsigned grid, hipStream_t s) {
    const MergeArgs& a = *(const MergeArgs*)args;
    if (a.ranked && a.k > 256) hipLaunchKernelGGL((k_merge_big<16>), dim3(grid), dim3(64), 0, s, a);
    else if (a.ranked && a.k > 64) hipLaunchKernelGGL((k_merge_big<4>), dim3(grid), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_merge, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_decode_list_side(const void* args, unsigned grid, hipStream_t s) {
    const DecodeArgs& a = *(const DecodeArgs*)args;
    hipLaunchKernelGGL(k_decode_list_side, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}
hipError_t ds2i_launch_decode_list(const void* args, unsigned grid, hipStream_t s) {
    const DecodeArgs& a = *(const DecodeArgs*)args;
    hipLaunchKernelGGL(k_decode_list, dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_selftest(const uint32_t* in, uint32_t* out, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest, dim3(blocks), dim3(64), 0, s, in, out);
    return hipGetLastError();
}

hipError_t ds2i_launch_selftest_bm25(const uint32_t* freqs, const float* norm_lens, float* out, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_bm25, dim3((n + 63) / 64), dim3(64), 0, s, freqs, norm_lens, out, n);
    return hipGetLastError();
}

hipError_t ds2i_launch_copy_seed(const uint32_t* queries, uint32_t n, uint32_t k, const float* seed_topk, const uint32_t* seed_len,
                                 const unsigned long long* seed_count, float* out_topk, uint32_t* out_len,
                                 unsigned long long* out_count, hipStream_t s) {
    CopySeedArgs a{queries, n, k, seed_topk, seed_len, seed_count, out_topk, out_len, out_count};
    hipLaunchKernelGGL(k_copy_seed, dim3(n < 1024 ? n : 1024), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ds2i_launch_calib_read(const uint32_t* base, unsigned long long ndw, uint32_t* out, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL(k_calib_read, dim3(grid), dim3(64), 0, s, base, ndw, out);
    return hipGetLastError();
}

size_t ds2i_sizeof_batch_args() { return sizeof(BatchArgs); }
size_t ds2i_sizeof_decode_args() { return sizeof(DecodeArgs); }
}
#endif // !DS2I_TU_TMAX
