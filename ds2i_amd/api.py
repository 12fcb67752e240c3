"""ctypes binding of libds2i_hip.so + a Python mirror of ds2i's Index / query-operator concepts.

Names follow the reference (queries.hpp): and_query, or_query, ranked_and_query, wand_query,
maxscore_query, ranked_or_query; an operator is called as op(index, terms) and ranked operators
expose topk(), exactly like queries.cpp:13-62 drives them. Batched entry points take a list of
queries (the batch boundary this framework inserts at queries.cpp:26).
"""
import ctypes as C
import os

import numpy as np

CODECS = {"block_optpfor": 0, "block_varint": 1, "block_interpolative": 2, "block_qmx": 3, "block_mixed": 4, "opt": 5,
          "ef": 6, "single": 7, "uniform": 8}
FREQ_INDEX_KINDS = ["opt", "ef", "single", "uniform"]  # freq_index<...> layouts (index_types.hpp:18-32)
BLOCK_CODECS = ["block_optpfor", "block_varint", "block_interpolative", "block_qmx", "block_mixed"]
OPS = {"and": 0, "and_freq": 1, "or": 2, "or_freq": 3, "ranked_and": 4, "wand": 5, "maxscore": 6, "ranked_or": 7}
REFERENCE_ORDER = 0x100
NO_COUNTERS = 0x200  # run the kernels compiled without the statistics counters (stats then carry kernel_ms only)
_RANKED = {4, 5, 6, 7}

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class Ds2iError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ds2i_hip error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("docs_blocks_decoded", C.c_uint64), ("freqs_blocks_decoded", C.c_uint64),
                ("block_max_examined", C.c_uint64), ("algorithmic_bytes", C.c_uint64), ("postings_scored", C.c_uint64),
                ("rounds", C.c_uint64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class GroupStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("lists", C.c_uint32), ("units", C.c_uint32), ("queries", C.c_uint32), ("pipelined_stream", C.c_int)]


def _class_groups(fn, handle, cls):
    n = C.c_uint32()
    _check(fn(handle, cls, None, 0, C.byref(n)))
    arr = (GroupStats * max(1, n.value))()
    _check(fn(handle, cls, arr, n.value, C.byref(n)))
    return [{"kernel_ms": g.kernel_ms, "lists": g.lists, "units": g.units, "queries": g.queries, "pipelined_stream": bool(g.pipelined_stream)}
            for g in arr[:n.value]]


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("num_docs", C.c_uint32), ("num_terms", C.c_uint32), ("zipf_exp", C.c_double),
                ("top_df_frac", C.c_double), ("min_len", C.c_uint32), ("clustered_every", C.c_uint32),
                ("topics", C.c_uint32), ("topic_boost", C.c_uint32)]  # correlated terms (0, 0: independent lists)


def library_path():
    # DS2I_LIB_VARIANT=name loads a diagnostic / A-B build of the same library (ds2i_amd/build.py, DS2I_BUILD_VARIANT)
    variant = os.environ.get("DS2I_LIB_VARIANT", "")
    if variant:
        return os.path.join(_HERE, "..", "profiles", "tmp_libs", "lib_%s.so" % variant)
    return os.path.join(_HERE, "libds2i_hip.so")


def lib():
    """Loads the shared library; raises loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError("libds2i_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the query path)")
        L = C.CDLL(path)
        vp, u32p, u64p, fp = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)
        L.ds2i_hip_last_error.restype = C.c_char_p
        L.ds2i_hip_set_option.argtypes = [C.c_char_p, C.c_char_p]
        L.ds2i_hip_index_open.argtypes = [C.c_int, C.c_int, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(vp)]
        L.ds2i_hip_index_close.argtypes = [vp]
        L.ds2i_hip_index_close.restype = None
        for f in ("ds2i_hip_index_size", "ds2i_hip_index_num_docs", "ds2i_hip_index_device_bytes"):
            getattr(L, f).argtypes = [vp]
            getattr(L, f).restype = C.c_uint64
        L.ds2i_hip_index_get_info.argtypes = [vp, vp]
        L.ds2i_hip_list_size.argtypes = [vp, C.c_uint32, u64p]
        L.ds2i_hip_decode_list.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint64, u64p]
        L.ds2i_hip_query_batch.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp, C.POINTER(Stats)]
        L.ds2i_hip_batch_prepare.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_uint32, C.c_int, C.POINTER(vp)]
        L.ds2i_hip_batch_run.argtypes = [vp, C.POINTER(Stats)]
        L.ds2i_hip_batch_class_stats.argtypes = [vp, C.c_int, C.POINTER(Stats), u32p]
        L.ds2i_hip_batch_phase_cycles.argtypes = [vp, C.c_int, vp, C.c_int]
        L.ds2i_hip_batch_class_groups.argtypes = [vp, C.c_int, vp, C.c_uint32, u32p]
        L.ds2i_hip_pipeline_class_groups.argtypes = [vp, C.c_int, vp, C.c_uint32, u32p]
        L.ds2i_hip_batch_fetch.argtypes = [vp, vp, vp, vp, vp]
        L.ds2i_hip_batch_match_total.argtypes = [vp, u64p]
        L.ds2i_hip_batch_fetch_matches.argtypes = [vp, vp, vp]
        L.ds2i_hip_batch_free.argtypes = [vp]
        L.ds2i_hip_batch_free.restype = None
        L.ds2i_hip_batch_set_instrumented.argtypes = [vp, C.c_int]
        L.ds2i_hip_pipeline_create.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
        L.ds2i_hip_pipeline_destroy.argtypes = [vp]
        L.ds2i_hip_pipeline_destroy.restype = None
        L.ds2i_hip_pipeline_submit.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_uint32, u64p]
        L.ds2i_hip_pipeline_wait.argtypes = [vp, C.c_uint64, vp, vp, vp, C.POINTER(Stats)]
        L.ds2i_hip_pipeline_set_instrumented.argtypes = [vp, C.c_int]
        L.ds2i_hip_pipeline_class_stats.argtypes = [vp, C.c_int, C.POINTER(Stats), u32p]
        L.ds2i_hip_batch_enable_block_profile.argtypes = [vp]
        L.ds2i_hip_batch_block_profile.argtypes = [vp, vp, C.c_uint64, u64p]
        L.ds2i_hybrid_default_model.argtypes = [vp]
        L.ds2i_hybrid_default_model.restype = None
        L.ds2i_hybrid_create.argtypes = [C.c_uint64, vp, C.POINTER(vp)]
        L.ds2i_hybrid_add_posting_list.argtypes = [vp, C.c_uint64, vp, vp, vp]
        L.ds2i_hybrid_analyse.argtypes = [vp, C.c_int, u64p, u64p]
        L.ds2i_hybrid_freeze.argtypes = [vp, C.c_uint64, C.c_int, C.POINTER(vp), C.POINTER(C.c_double), u64p,
                                         C.POINTER(C.c_double), u64p]
        L.ds2i_synth_build_hybrid.argtypes = [vp, C.c_int, vp, vp, C.c_double, C.POINTER(vp), C.POINTER(vp), u64p, u64p]
        L.ds2i_hybrid_free.argtypes = [vp]
        L.ds2i_hybrid_free.restype = None
        L.ds2i_hip_calibration_read.argtypes = [vp, u64p]
        L.ds2i_hip_list_block_weights.argtypes = [vp, C.c_uint32, vp, C.c_uint64, u64p]
        L.ds2i_hip_list_range_table.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, u64p, u32p, fp]
        L.ds2i_hip_selftest_scan.argtypes = [C.c_int, vp, vp, C.c_uint32]
        L.ds2i_hip_selftest_bm25.argtypes = [C.c_int, vp, vp, vp, C.c_uint32]
        L.ds2i_hip_synth_encode.argtypes = [C.c_int, C.POINTER(SynthParams), C.c_int, C.POINTER(vp), C.POINTER(vp), u64p,
                                            C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ds2i_hip_encode_index.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint64, vp, vp, vp, C.POINTER(vp), C.POINTER(C.c_double)]
        # build side
        L.ds2i_blob_data.argtypes = [vp]
        L.ds2i_blob_data.restype = vp
        L.ds2i_blob_size.argtypes = [vp]
        L.ds2i_blob_size.restype = C.c_size_t
        L.ds2i_blob_free.argtypes = [vp]
        L.ds2i_blob_free.restype = None
        L.ds2i_builder_create.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp)]
        L.ds2i_builder_add_posting_list.argtypes = [vp, C.c_uint64, vp, vp]
        L.ds2i_builder_freeze.argtypes = [vp, C.POINTER(vp)]
        L.ds2i_builder_free.argtypes = [vp]
        L.ds2i_builder_free.restype = None
        L.ds2i_wand_create.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
        L.ds2i_wand_add_list.argtypes = [vp, C.c_uint64, vp, vp]
        L.ds2i_wand_freeze.argtypes = [vp, C.POINTER(vp)]
        L.ds2i_wand_free.argtypes = [vp]
        L.ds2i_wand_free.restype = None
        L.ds2i_encode_block.argtypes = [C.c_int, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.ds2i_encode_vbyte.argtypes = [C.c_uint32, C.POINTER(vp)]
        L.ds2i_write_sequence.argtypes = [C.c_int, vp, C.c_uint64, C.c_uint64, vp, C.POINTER(vp), u64p]
        L.ds2i_encode_posting_list.argtypes = [C.c_int, C.c_uint32, vp, vp, C.POINTER(vp)]
        L.ds2i_opt_list_directory.argtypes = [vp, C.c_size_t, C.c_uint32, C.POINTER(vp), C.POINTER(vp), u64p]
        L.ds2i_freq_list_directory.argtypes = [C.c_int, vp, C.c_size_t, C.c_uint32, C.POINTER(vp), C.POINTER(vp), u64p]
        L.ds2i_synth_list_upper_bound.argtypes = [C.POINTER(SynthParams), C.c_uint32]
        L.ds2i_synth_list_upper_bound.restype = C.c_uint64
        L.ds2i_synth_list.argtypes = [C.POINTER(SynthParams), C.c_uint32, vp, vp, C.c_uint64, u64p]
        L.ds2i_synth_doc_sizes.argtypes = [C.POINTER(SynthParams), vp]
        L.ds2i_synth_queries.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, vp, vp]
        L.ds2i_synth_queries_topical.argtypes = [C.POINTER(SynthParams), C.c_uint64, C.c_uint32, C.c_uint32, vp, vp]
        L.ds2i_synth_build.argtypes = [C.POINTER(SynthParams), C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), u64p]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise Ds2iError(rc, lib().ds2i_hip_last_error().decode("utf-8", "replace"))


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _take_blob(handle):
    L = lib()
    n = L.ds2i_blob_size(handle)
    # (c_char * n).raw handles images > 2 GiB (string_at takes a C int size)
    out = (C.c_char * n).from_address(L.ds2i_blob_data(handle)).raw if n else b""
    L.ds2i_blob_free(handle)
    return out


def _codec(c):
    return CODECS[c] if isinstance(c, str) else int(c)


# ---------------------------------------------------------------- build side (host, CPU)
def encode_block(codec, values, sum_of_values=0xFFFFFFFF):
    v = _u32(values)
    h = C.c_void_p()
    _check(lib().ds2i_encode_block(_codec(codec), _ptr(v), C.c_uint32(sum_of_values & 0xFFFFFFFF), len(v), C.byref(h)))
    return _take_blob(h)


def encode_vbyte(value):
    h = C.c_void_p()
    _check(lib().ds2i_encode_vbyte(value, C.byref(h)))
    return _take_blob(h)


SEQUENCE_KINDS = ("elias_fano", "ranked_bitvector", "indexed", "strict", "partitioned_indexed", "partitioned_strict",
                  "uniform_indexed", "uniform_strict")


def write_sequence(seq_kind, values, universe, params=None):
    """One sequence of the Elias-Fano family written into a fresh bit string (ds2i_write_sequence).
    Returns (u64 words, nbits). params = (ef_log_sampling0, ef_log_sampling1, rb_log_rank1_sampling,
    rb_log_sampling1, log_partition_size) or None for global_parameters' defaults."""
    v = np.ascontiguousarray(values, dtype=np.uint64)
    h = C.c_void_p()
    nbits = C.c_uint64()
    pp = None
    if params is not None:
        pp = np.asarray(params, dtype=np.uint8)
        assert pp.shape == (5,)
    _check(lib().ds2i_write_sequence(SEQUENCE_KINDS.index(seq_kind), _ptr(v), len(v), int(universe),
                                     _ptr(pp) if pp is not None else None, C.byref(h), C.byref(nbits)))
    return np.frombuffer(_take_blob(h), dtype=np.uint64), int(nbits.value)


def encode_posting_list(codec, docs, freqs):
    d, f = _u32(docs), _u32(freqs)
    h = C.c_void_p()
    _check(lib().ds2i_encode_posting_list(_codec(codec), len(d), _ptr(d), _ptr(f), C.byref(h)))
    return _take_blob(h)


def build_index(codec, num_docs, lists):
    """lists: iterable of (docs, freqs). Returns the frozen block_freq_index image (bytes)."""
    L = lib()
    b = C.c_void_p()
    _check(L.ds2i_builder_create(_codec(codec), num_docs, C.byref(b)))
    try:
        for docs, freqs in lists:
            d, f = _u32(docs), _u32(freqs)
            _check(L.ds2i_builder_add_posting_list(b, len(d), _ptr(d), _ptr(f)))
        h = C.c_void_p()
        _check(L.ds2i_builder_freeze(b, C.byref(h)))
        return _take_blob(h)
    finally:
        L.ds2i_builder_free(b)


def gpu_encode_index(num_docs, lists, device=0, codec="block_optpfor"):
    """The index image of `lists` (iterable of (docs, freqs)) encoded ON THE GPU (ds2i_hip_encode_index): byte-identical
    to build_index(codec, num_docs, lists). Returns (image bytes, device milliseconds of the two kernel passes)."""
    lists = [(_u32(dd), _u32(ff)) for dd, ff in lists]
    offs = np.zeros(len(lists) + 1, dtype=np.uint64)
    for i, (dd, _) in enumerate(lists):
        offs[i + 1] = offs[i] + len(dd)
    docs = np.concatenate([dd for dd, _ in lists]) if lists else np.zeros(1, np.uint32)
    freqs = np.concatenate([ff for _, ff in lists]) if lists else np.zeros(1, np.uint32)
    h, ms = C.c_void_p(), C.c_double()
    _check(lib().ds2i_hip_encode_index(device, _codec(codec), num_docs, len(lists), _ptr(offs), _ptr(docs), _ptr(freqs),
                                       C.byref(h), C.byref(ms)))
    return _take_blob(h), ms.value


def synth_build_gpu(p, device=0, threads=0):
    """The synthetic collection as a block_optpfor index encoded on the GPU (ds2i_hip_synth_encode).
    -> (index image, wand image, postings, dict(generate_s, device_ms))"""
    hi, hw = C.c_void_p(), C.c_void_p()
    n, gs, ms = C.c_uint64(), C.c_double(), C.c_double()
    _check(lib().ds2i_hip_synth_encode(device, C.byref(p), threads, C.byref(hi), C.byref(hw), C.byref(n), C.byref(gs), C.byref(ms)))
    return _take_blob(hi), _take_blob(hw), n.value, {"generate_s": gs.value, "device_ms": ms.value}


def build_wand(doc_sizes, lists):
    L = lib()
    s = _u32(doc_sizes)
    w = C.c_void_p()
    _check(L.ds2i_wand_create(_ptr(s), len(s), C.byref(w)))
    try:
        for docs, freqs in lists:
            d, f = _u32(docs), _u32(freqs)
            _check(L.ds2i_wand_add_list(w, len(d), _ptr(d), _ptr(f)))
        h = C.c_void_p()
        _check(L.ds2i_wand_freeze(w, C.byref(h)))
        return _take_blob(h)
    finally:
        L.ds2i_wand_free(w)


class HybridModel(C.Structure):
    """MI355X decode cost per 128-value block, wave-level instructions (ds2i_hybrid_model)."""
    _fields_ = [(n, C.c_float) for n in ("pfor_base", "pfor_exc", "pfor_exc_many", "varint", "interp_base", "interp_node")]

    @classmethod
    def default(cls):
        m = cls()
        lib().ds2i_hybrid_default_model(C.byref(m))
        return m


class HybridBuilder:
    """block_mixed space/time optimiser (ds2i_hybrid_*; reference optimal_hybrid_index.cpp)."""

    def __init__(self, num_docs, model=None):
        self._h = C.c_void_p()
        _check(lib().ds2i_hybrid_create(num_docs, C.byref(model) if model is not None else None, C.byref(self._h)))

    def add_posting_list(self, docs, freqs, access=None):
        d = np.ascontiguousarray(docs, dtype=np.uint32)
        f = np.ascontiguousarray(freqs, dtype=np.uint32)
        a = None
        if access is not None:
            a = np.ascontiguousarray(access, dtype=np.uint32).reshape(-1)
            assert a.size == 2 * ((len(d) + 127) // 128)
        _check(lib().ds2i_hybrid_add_posting_list(self._h, len(d), _ptr(d), _ptr(f), _ptr(a) if a is not None else None))

    def analyse(self, threads=0):
        lo, hi = C.c_uint64(), C.c_uint64()
        _check(lib().ds2i_hybrid_analyse(self._h, threads, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def freeze(self, budget_bytes=None, threads=0):
        """-> (image bytes, dict(rate, space, model_time, type_counts))"""
        h = C.c_void_p()
        rate, t, space = C.c_double(), C.c_double(), C.c_uint64()
        tc = (C.c_uint64 * 6)()
        budget = 0xFFFFFFFFFFFFFFFF if budget_bytes is None else int(budget_bytes)
        _check(lib().ds2i_hybrid_freeze(self._h, budget, threads, C.byref(h), C.byref(rate), C.byref(space), C.byref(t), tc))
        return _take_blob(h), {"rate": rate.value, "space": space.value, "model_time": t.value,
                               "type_counts": {"docs": list(tc[0:3]), "freqs": list(tc[3:6])}}

    def close(self):
        if self._h:
            lib().ds2i_hybrid_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def synth_build_hybrid(p, budget_frac=0.5, threads=0, access=None, model=None):
    """The synthetic collection as an optimised block_mixed index -> (index image, wand image, postings, type_counts)."""
    hi, hw = C.c_void_p(), C.c_void_p()
    n = C.c_uint64()
    tc = (C.c_uint64 * 6)()
    a = None if access is None else np.ascontiguousarray(access, dtype=np.uint32).reshape(-1)
    _check(lib().ds2i_synth_build_hybrid(C.byref(p), threads, C.byref(model) if model is not None else None,
                                         _ptr(a) if a is not None else None, float(budget_frac), C.byref(hi), C.byref(hw),
                                         C.byref(n), tc))
    return _take_blob(hi), _take_blob(hw), n.value, {"docs": list(tc[0:3]), "freqs": list(tc[3:6])}


def opt_list_directory(image, term, kind="opt"):
    """The chunk directory the GPU upload builds for one list of a freq_index image (inspection / tests)."""
    hc, hk = C.c_void_p(), C.c_void_p()
    info = (C.c_uint64 * 5)()
    _check(lib().ds2i_freq_list_directory(_codec(kind), image, len(image), term, C.byref(hc), C.byref(hk), info))
    cmax = np.frombuffer(_take_blob(hc), dtype=np.uint32)
    chunks = np.frombuffer(_take_blob(hk), dtype=np.uint32).reshape(-1, 12)
    return cmax, chunks, [int(x) for x in info]


def synth_list(p, term):
    cap = int(p.num_docs)
    d = np.empty(cap, dtype=np.uint32)
    f = np.empty(cap, dtype=np.uint32)
    n = C.c_uint64()
    _check(lib().ds2i_synth_list(C.byref(p), term, _ptr(d), _ptr(f), cap, C.byref(n)))
    return d[:n.value].copy(), f[:n.value].copy()


def synth_doc_sizes(p):
    s = np.empty(int(p.num_docs), dtype=np.uint32)
    _check(lib().ds2i_synth_doc_sizes(C.byref(p), _ptr(s)))
    return s


def set_option(name, value):
    """a DS2I_* tuning knob without the environment (ds2i_hip_set_option): before the first batch is planned"""
    _check(lib().ds2i_hip_set_option(name.encode(), None if value is None else str(value).encode()))


def synth_queries(seed, num_terms, nq):
    t = np.empty(11 * nq + 1, dtype=np.uint32)
    o = np.empty(nq + 1, dtype=np.uint32)
    _check(lib().ds2i_synth_queries(seed, num_terms, nq, _ptr(t), _ptr(o)))
    return [t[o[i]:o[i + 1]].tolist() for i in range(nq)]


def synth_queries_topical(p, seed, nq, same_topic_pct=25):
    """synth_queries over a correlated collection (p.topics > 1): same_topic_pct percent of the multi-term queries take
    all their terms from one topic"""
    t = np.empty(11 * nq + 1, dtype=np.uint32)
    o = np.empty(nq + 1, dtype=np.uint32)
    _check(lib().ds2i_synth_queries_topical(C.byref(p), seed, nq, same_topic_pct, _ptr(t), _ptr(o)))
    return [t[o[i]:o[i + 1]].tolist() for i in range(nq)]


def synth_build(p, codec, threads=0):
    """Returns (index_image, wand_image, total_postings) for the synthetic collection p."""
    hi, hw = C.c_void_p(), C.c_void_p()
    tot = C.c_uint64()
    _check(lib().ds2i_synth_build(C.byref(p), _codec(codec), threads, C.byref(hi), C.byref(hw), C.byref(tot)))
    return _take_blob(hi), _take_blob(hw), tot.value


# ---------------------------------------------------------------- query side (GPU)
def _flatten(queries):
    offs = np.zeros(len(queries) + 1, dtype=np.uint32)
    for i, q in enumerate(queries):
        offs[i + 1] = offs[i] + len(q)
    terms = np.zeros(max(int(offs[-1]), 1), dtype=np.uint32)
    pos = 0
    for q in queries:
        terms[pos:pos + len(q)] = q
        pos += len(q)
    return terms, offs


def _op(op):
    return OPS[op] if isinstance(op, str) else int(op)


class Batch:
    """A prepared query batch resident in HBM (ds2i_hip_batch_*)."""

    def __init__(self, index, op, queries, k=10, want_matches=False, reference_order=False):
        self.index, self.nq, self.k = index, len(queries), k
        self.op = _op(op) | (REFERENCE_ORDER if reference_order else 0)
        terms, offs = _flatten(queries)
        self._h = C.c_void_p()
        _check(lib().ds2i_hip_batch_prepare(index._h, self.op, k, _ptr(terms), _ptr(offs), self.nq,
                                            1 if want_matches else 0, C.byref(self._h)))
        self.k = k if (self.op & 0xFF) in _RANKED else max(k, 1)

    def set_instrumented(self, on):
        """Statistics counters on (default) / off -- see ds2i_hip_batch_set_instrumented."""
        _check(lib().ds2i_hip_batch_set_instrumented(self._h, 1 if on else 0))

    def enable_block_profile(self):
        """Start counting per-block decodes (block indexes; instrumented runs accumulate)."""
        _check(lib().ds2i_hip_batch_enable_block_profile(self._h))

    def block_profile(self):
        """uint32[total_blocks, 2]: (docs decodes, freqs decodes) per block, lists in index order."""
        tot = C.c_uint64()
        _check(lib().ds2i_hip_batch_block_profile(self._h, None, 0, C.byref(tot)))
        out = np.zeros((max(tot.value, 1), 2), dtype=np.uint32)
        _check(lib().ds2i_hip_batch_block_profile(self._h, _ptr(out), out.size, C.byref(tot)))
        return out[:tot.value]

    def run(self):
        st = Stats()
        _check(lib().ds2i_hip_batch_run(self._h, C.byref(st)))
        return st

    def class_stats(self, cls):
        st, n = Stats(), C.c_uint32()
        _check(lib().ds2i_hip_batch_class_stats(self._h, cls, C.byref(st), C.byref(n)))
        return st, n.value

    def class_groups(self, cls):
        """launch groups of kernel class `cls` in the last run (ds2i_hip_batch_class_groups)"""
        return _class_groups(lib().ds2i_hip_batch_class_groups, self._h, cls)

    def phase_cycles(self, cls):
        names = ("total", "docs", "freqs", "find", "member", "score", "topk", "prolog", "probe", "insert", "stream", "prefetch", "floor", "unit",
                 "n_visit", "n_surv1", "n_surv2", "n_bdocs", "n_bfreqs", "n_heap", "n_liverounds", "n_alive", "n_gblocks")
        out = np.zeros(len(names), dtype=np.uint64)
        _check(lib().ds2i_hip_batch_phase_cycles(self._h, cls, _ptr(out), len(names)))
        return dict(zip(names, out.tolist()))

    def fetch(self):
        nq = max(self.nq, 1)
        count = np.zeros(nq, dtype=np.uint64)
        topk = np.full((nq, self.k), -np.inf, dtype=np.float32)
        tlen = np.zeros(nq, dtype=np.uint32)
        fsum = np.zeros(nq, dtype=np.uint64)
        _check(lib().ds2i_hip_batch_fetch(self._h, _ptr(count), _ptr(topk), _ptr(tlen), _ptr(fsum)))
        return count[:self.nq], topk[:self.nq], tlen[:self.nq], fsum[:self.nq]

    def fetch_matches(self, counts):
        tot = C.c_uint64()
        _check(lib().ds2i_hip_batch_match_total(self._h, C.byref(tot)))
        offs = np.zeros(self.nq + 1, dtype=np.uint64)
        m = np.zeros(max(tot.value, 1), dtype=np.uint32)
        _check(lib().ds2i_hip_batch_fetch_matches(self._h, _ptr(offs), _ptr(m)))
        return [m[int(offs[i]):int(offs[i]) + int(counts[i])].copy() for i in range(self.nq)]

    def close(self):
        if self._h:
            lib().ds2i_hip_batch_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """Pipelined submit / wait over reusable batch slots (ds2i_hip_pipeline_*): submit() plans the batch on the host
    and enqueues upload + kernels + result copy without waiting for the device, so the next submit() overlaps with
    the kernels of this one. A serving loop keeps `depth` batches in flight."""

    def __init__(self, index, depth=3):
        self.index, self.depth = index, depth
        self._h = C.c_void_p()
        self._meta = {}
        _check(lib().ds2i_hip_pipeline_create(index._h, depth, C.byref(self._h)))

    def set_instrumented(self, on):
        _check(lib().ds2i_hip_pipeline_set_instrumented(self._h, 1 if on else 0))

    def submit(self, op, queries, k=10):
        """queries: list of term lists, or an already flattened (terms uint32[], offsets uint32[nq+1]) pair."""
        terms, offs = queries if isinstance(queries, tuple) else _flatten(queries)
        nq = len(offs) - 1
        t = C.c_uint64()
        _check(lib().ds2i_hip_pipeline_submit(self._h, _op(op), k, _ptr(terms), _ptr(offs), nq, C.byref(t)))
        self._meta[t.value] = (nq, k if (_op(op) & 0xFF) in _RANKED else 1)
        return t.value

    def wait(self, ticket, stats=False):
        nq, k = self._meta.pop(ticket, (0, 1))
        count = np.zeros(max(nq, 1), dtype=np.uint64)
        topk = np.full((max(nq, 1), k), -np.inf, dtype=np.float32)
        tlen = np.zeros(max(nq, 1), dtype=np.uint32)
        st = Stats()
        _check(lib().ds2i_hip_pipeline_wait(self._h, ticket, _ptr(count), _ptr(topk), _ptr(tlen), C.byref(st) if stats else None))
        return (count[:nq], topk[:nq], tlen[:nq], st) if stats else (count[:nq], topk[:nq], tlen[:nq])

    def class_stats(self, cls):
        """(Stats, queries) of kernel class `cls` for the ticket collected last."""
        st, n = Stats(), C.c_uint32()
        _check(lib().ds2i_hip_pipeline_class_stats(self._h, cls, C.byref(st), C.byref(n)))
        return st, n.value

    def class_groups(self, cls):
        """launch groups of kernel class `cls` for the ticket collected last (ds2i_hip_pipeline_class_groups)"""
        return _class_groups(lib().ds2i_hip_pipeline_class_groups, self._h, cls)

    def close(self):
        if self._h:
            lib().ds2i_hip_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def flatten_queries(queries):
    """(terms, offsets) arrays of a query batch -- the form that crosses the C ABI."""
    return _flatten(queries)


class IndexInfo(C.Structure):
    """ds2i_hip_index_info (include/ds2i_hip.h)"""
    _fields_ = [("index_bytes", C.c_uint64), ("skip_table_bytes", C.c_uint64), ("block_weight_bytes", C.c_uint64), ("range_table_bytes", C.c_uint64),
                ("norm_len_bytes", C.c_uint64), ("total_blocks", C.c_uint64), ("total_postings", C.c_uint64), ("has_block_weights", C.c_int),
                ("has_range_tables", C.c_int), ("has_bitmaps", C.c_int), ("has_membership_hints", C.c_int), ("range_table_entries_per_posting", C.c_int),
                ("side_table_bytes", C.c_uint64), ("has_side_tables", C.c_int), ("transcoded_from", C.c_int), ("table_budget_bytes", C.c_uint64)]


class Index:
    """block_freq_index resident in one GPU's HBM (Index concept: size(), num_docs(), operator[])."""

    def __init__(self, kind, index_image, wand_image=None, device=0):
        self._h = C.c_void_p()
        self.kind = _codec(kind)
        wi = wand_image if wand_image is not None else None
        _check(lib().ds2i_hip_index_open(device, self.kind, index_image, len(index_image), wi,
                                         len(wand_image) if wand_image is not None else 0, C.byref(self._h)))

    def size(self):
        return lib().ds2i_hip_index_size(self._h)

    def num_docs(self):
        return lib().ds2i_hip_index_num_docs(self._h)

    def device_bytes(self):
        return lib().ds2i_hip_index_device_bytes(self._h)

    def info(self):
        """what the upload put into HBM beside the index image (ds2i_hip_index_get_info), as a dict"""
        i = IndexInfo()
        _check(lib().ds2i_hip_index_get_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in IndexInfo._fields_}

    def list_size(self, term):
        n = C.c_uint64()
        _check(lib().ds2i_hip_list_size(self._h, term, C.byref(n)))
        return n.value

    def __getitem__(self, term):
        """index[term]: the whole list enumerated on the GPU -> (docs, freqs)."""
        n = self.list_size(term)
        d = np.empty(n, dtype=np.uint32)
        f = np.empty(n, dtype=np.uint32)
        got = C.c_uint64()
        _check(lib().ds2i_hip_decode_list(self._h, term, _ptr(d), _ptr(f), n, C.byref(got)))
        return d, f

    def calibration_read(self):
        n = C.c_uint64()
        _check(lib().ds2i_hip_calibration_read(self._h, C.byref(n)))
        return n.value

    def block_weights(self, term):
        """bmw[] of one list (ds2i_hip_list_block_weights): float32 per block, or an empty array without wand data"""
        n = C.c_uint64()
        out = np.zeros(int(self.list_size(term)) + 8, dtype=np.float32)  # (a chunk holds at least one posting)
        _check(lib().ds2i_hip_list_block_weights(self._h, term, _ptr(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    def range_table(self, term, level=1):
        """(bytes, shift, list_max) of one level of one list's doc-id-range table (ds2i_hip_list_range_table)"""
        n, sh, mx = C.c_uint64(), C.c_uint32(), C.c_float()
        out = np.zeros(int(self.num_docs()) + 64, dtype=np.uint8)
        _check(lib().ds2i_hip_list_range_table(self._h, term, level, _ptr(out), len(out), C.byref(n), C.byref(sh), C.byref(mx)))
        return out[:n.value].copy(), int(sh.value), float(mx.value)

    def query_batch(self, op, queries, k=10):
        terms, offs = queries if isinstance(queries, tuple) else _flatten(queries)
        nq = len(offs) - 1
        count = np.zeros(max(nq, 1), dtype=np.uint64)
        topk = np.full((max(nq, 1), max(k, 1)), -np.inf, dtype=np.float32)
        tlen = np.zeros(max(nq, 1), dtype=np.uint32)
        st = Stats()
        _check(lib().ds2i_hip_query_batch(self._h, _op(op), k, _ptr(terms), _ptr(offs), nq, _ptr(count), _ptr(topk),
                                          _ptr(tlen), C.byref(st)))
        return count[:nq], topk[:nq], tlen[:nq], st

    def close(self):
        if self._h:
            lib().ds2i_hip_index_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- query-operator concept
class _query_op:
    op = None
    ranked = False

    def __init__(self, wdata=None, k=10):
        self.k = k
        self._topk = []

    def __call__(self, index, terms):
        """uint64_t operator()(Index const&, term_id_vec) -- a batch of one query."""
        return int(self.batch(index, [list(terms)])[0])

    def batch(self, index, queries):
        count, topk, tlen, _ = index.query_batch(self.op, queries, self.k)
        if self.ranked:
            self._topk = [topk[i, :tlen[i]].copy() for i in range(len(queries))]
        return count

    def topk(self, i=-1):
        return self._topk[i]


class and_query(_query_op):
    def __init__(self, with_freqs=False):
        super().__init__()
        self.op = "and_freq" if with_freqs else "and"


class or_query(_query_op):
    def __init__(self, with_freqs=False):
        super().__init__()
        self.op = "or_freq" if with_freqs else "or"


class ranked_and_query(_query_op):
    op, ranked = "ranked_and", True


class wand_query(_query_op):
    op, ranked = "wand", True


class maxscore_query(_query_op):
    op, ranked = "maxscore", True


class ranked_or_query(_query_op):
    op, ranked = "ranked_or", True
