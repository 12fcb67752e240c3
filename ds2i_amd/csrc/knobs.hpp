// Every environment knob of libds2i_hip.so, read in ONE place (capi.cpp: ds2i_knobs). Twenty of them: what an upload builds, how a
// batch is cut and which kernel family answers it, diagnostics. (Re-)read by every ds2i_hip_index_open; they hold for that index and for
// the batches planned until the next upload. ds2i_hip_set_option(name, value) sets one without the environment.
// (Rounds 2-5 accumulated fifty A/B switches; the ones whose alternative lost twice are gone together with what only they reached --
// CHANGELOG.md has the measurements.)
#pragma once
#include <cstdint>

struct Ds2iKnobs {
    // ---- upload (ds2i_hip_index_open)
    double rmw_g;        // DS2I_RMW_G: range-table entries per posting, a power of two (default 4; 2 = 10 GB less at GOV2 scale, -4 % ranked_and, -8 % wand)
    bool rmw_g_set;      //   ... set explicitly (a DS2I_TABLE_BUDGET then does not override it)
    bool no_rmw;         // DS2I_NO_RMW: no doc-id-range tables (ranked kernels fall back to block-max pruning only; wand to k_disjunctive)
    bool no_rmh;         // DS2I_NO_RMH: no membership hints
    bool no_bitmaps;     // DS2I_NO_BITMAPS: no exact bitmaps of the dense lists
    bool no_bmw;         // DS2I_NO_BMW: no block-max weights (and none of the tables built from them)
    bool no_xslots;      // DS2I_NO_XSLOTS: no exception side slots / tail table (the kernels parse the on-disk OptPFor blocks)
    bool rmw_require;    // DS2I_RMW_REQUIRE: an upload that cannot afford its tables fails (DS2I_ENOMEM) instead of running without
    bool mixed_native;   // DS2I_MIXED_NATIVE: block_mixed images are queried as they are (no transcoding at upload)
    bool pef_native;     // DS2I_PEF_NATIVE: opt / ef / single / uniform images are queried as they are
    char table_budget[32];    // DS2I_TABLE_BUDGET: "<bytes>" or "<factor>x" (of the caller's image); empty = none
    // ---- batches (planner / launcher)
    unsigned plan_threads;    // DS2I_PLAN_THREADS: host threads planning a batch (0 = default: up to 4, the process's CPU share)
    double unit_factor;       // DS2I_UNIT_FACTOR: work units per resident wave (0 = default per operator)
    uint32_t unit_cap;        // DS2I_UNIT_CAP: at most this many blocks of the shortest list per unit of a ranked conjunction (0 = off)
    uint32_t ut_blocks;       // DS2I_UT_BLOCKS: blocks of the driving list per unit of wand / maxscore / ranked_or (default 320)
    uint32_t stream_nt_max;   // DS2I_STREAM_NT_MAX: ranked_and / and queries of up to this many lists run k_ranked_stream (default 16)
    bool no_ranked_stream;    // DS2I_NO_RANKED_STREAM: ranked_and / and through the class kernels (k_conjunctive)
    bool no_union_rstream;    // DS2I_NO_UNION_RSTREAM: wand / maxscore / ranked_or through k_union_topk instead of k_union_stream
    bool no_list_streams;     // DS2I_NO_LIST_STREAMS: no k_freq_stream (or_freq) / k_and_stream (and / and_freq of one-term and all-dense queries)
    bool decode_general;      // DS2I_DECODE_GENERAL: ds2i_hip_decode_list through the on-disk decoders although side slots exist
    bool unit_clock;          // DS2I_UNIT_CLOCK: instrumented runs record every unit's start / end and print where a class's time went
};
Ds2iKnobs ds2i_knobs(); // (a copy: an upload on another thread may be re-reading them)
