"""ORACLE -- test infrastructure only (see oracle/oracle.cpp header).

ctypes access to liboracle.so (CPU restatement of the reference's read path) and, when built,
oracle/_ref/libqmx_ref.so (the reference's own QMX codec). Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package. The product never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
OPS = {"and": 0, "and_freq": 1, "or": 2, "or_freq": 3, "ranked_and": 4, "wand": 5, "maxscore": 6, "ranked_or": 7}
CODECS = {"block_optpfor": 0, "block_varint": 1, "block_interpolative": 2, "block_qmx": 3, "block_mixed": 4, "opt": 5,
          "ef": 6, "single": 7, "uniform": 8}
_lib = None
_ref = None


class Profile(C.Structure):
    _fields_ = [("docs_blocks", C.c_uint64), ("freqs_blocks", C.c_uint64), ("block_max_examined", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("postings_scored", C.c_uint64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def build(native=False, out=None):
    """Compiles liboracle.so (and oracle/_ref when /root/reference exists). Returns the .so path."""
    if native:
        out = out or os.path.join(_HERE, "liboracle_native.so")
        subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
                               "-o", out, os.path.join(_HERE, "oracle.cpp")])
        return out
    subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return os.path.join(_HERE, "liboracle.so")


def lib(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    if path is None and (not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(src)):
        build()
    L = C.CDLL(p)
    vp = C.c_void_p
    L.oracle_decode_block.argtypes = [C.c_int, vp, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_uint64)]
    L.oracle_decode_vbyte.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.oracle_qmx_decode_stream.argtypes = [vp, vp, C.c_uint64]
    L.oracle_bm25_doc_term_weight.argtypes = [vp, vp, C.c_uint64, vp]
    L.oracle_bm25_doc_term_weight.restype = None
    L.oracle_bm25_query_term_weight.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp]
    L.oracle_bm25_query_term_weight.restype = None
    L.oracle_qmx_decode_stream.restype = None
    L.oracle_index_open.argtypes = [C.c_int, vp, C.c_uint64, vp, C.c_uint64]
    L.oracle_index_open.restype = vp
    L.oracle_index_close.argtypes = [vp]
    L.oracle_index_close.restype = None
    for f in ("oracle_index_size", "oracle_index_num_docs"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_uint64
    L.oracle_list_offset.argtypes = [vp, C.c_uint64]
    L.oracle_list_offset.restype = C.c_uint64
    L.oracle_list_size.argtypes = [vp, C.c_uint64]
    L.oracle_list_size.restype = C.c_int64
    L.oracle_list_enumerate.argtypes = [vp, C.c_uint64, vp, vp, C.c_uint64]
    L.oracle_list_enumerate.restype = C.c_int64
    L.oracle_list_next_geq.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, vp]
    L.oracle_list_move.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, vp]
    L.oracle_sequence_selftest.argtypes = [C.c_int, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_uint64, vp]
    L.oracle_query.argtypes = [vp, C.c_int, C.c_uint32, vp, C.c_uint32, vp, vp, vp, C.c_uint64, vp, C.POINTER(Profile)]
    L.oracle_query.restype = C.c_int64
    L.oracle_query_batch.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp, vp, C.POINTER(Profile)]
    L.oracle_query_batch_mt.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp]
    L.oracle_perftest.argtypes = [vp, C.c_int, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp]
    if path is None:
        _lib = L
    return L


def sequence_selftest(kind, words, nbits, universe, seq, params=(9, 8, 9, 8, 7)):
    """The reference's sequence tests (test_generic_sequence.hpp:28-164, test_partitioned_sequence.cpp:13-44) with the
    oracle's enumerators reading `words` (u64 bit string). kind = index into ds2i_sequence_kind. Returns 0 or the code of
    the first failed requirement."""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    v = np.ascontiguousarray(seq, dtype=np.uint64)
    pp = np.asarray(params, dtype=np.uint8)
    return int(lib().oracle_sequence_selftest(int(kind), _p(w), w.nbytes, int(nbits), int(universe), _p(v), len(v), _p(pp)))


def bm25_doc_term_weight(freq, norm_len):
    """oracle.cpp's bm25::doc_term_weight, element-wise (float32)."""
    f = np.ascontiguousarray(freq, dtype=np.uint64)
    nl = np.ascontiguousarray(norm_len, dtype=np.float32)
    out = np.zeros(len(f), dtype=np.float32)
    lib().oracle_bm25_doc_term_weight(_p(f), _p(nl), len(f), _p(out))
    return out


def bm25_query_term_weight(qtf, df, num_docs):
    q = np.ascontiguousarray(qtf, dtype=np.uint64)
    d = np.ascontiguousarray(df, dtype=np.uint64)
    out = np.zeros(len(q), dtype=np.float32)
    lib().oracle_bm25_query_term_weight(_p(q), _p(d), int(num_docs), len(q), _p(out))
    return out


_ref_bm25 = None


def ref_bm25():
    """The reference's own bm25.hpp (oracle/_ref/libbm25_ref.so) or None when it was never built."""
    global _ref_bm25
    if _ref_bm25 is None:
        p = os.path.join(_HERE, "_ref", "libbm25_ref.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_bm25_doc_term_weight.argtypes = [C.c_uint64, C.c_float]
        R.ref_bm25_doc_term_weight.restype = C.c_float
        R.ref_bm25_query_term_weight.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        R.ref_bm25_query_term_weight.restype = C.c_float
        _ref_bm25 = R
    return _ref_bm25


def ref_qmx():
    """The reference's own QMX codec (oracle/_ref/libqmx_ref.so) or None when it was never built."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libqmx_ref.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_qmx_encode.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_qmx_encode.restype = C.c_size_t
        R.ref_qmx_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        R.ref_qmx_decode.restype = None
        _ref = R
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _op(op):
    return OPS[op] if isinstance(op, str) else int(op)


def _codec(c):
    return CODECS[c] if isinstance(c, str) else int(c)


def decode_block(codec, data, n, sum_of_values=0xFFFFFFFF):
    """-> (values[n], consumed_bytes). data is padded so the decoders' benign over-reads stay in bounds."""
    buf = np.frombuffer(bytes(data) + b"\0" * 64, dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint32)
    consumed = C.c_uint64()
    rc = lib().oracle_decode_block(_codec(codec), _p(buf), C.c_uint32(sum_of_values & 0xFFFFFFFF), n, _p(out), C.byref(consumed))
    if rc:
        raise RuntimeError("oracle_decode_block failed")
    return out, consumed.value


def decode_vbyte(data):
    buf = np.frombuffer(bytes(data) + b"\0" * 8, dtype=np.uint8)
    v = C.c_uint32()
    n = lib().oracle_decode_vbyte(_p(buf), C.byref(v))
    return v.value, n


def _padded(data, pad):
    """one zero-padded copy of an image (the decoders' benign over-reads must stay in bounds)"""
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros(len(src) + pad, dtype=np.uint8)
    out[:len(src)] = src
    return out


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


class Index:
    """The reference path on the CPU: block_freq_index + wand_data + query functors."""

    def __init__(self, kind, index_image, wand_image=None, libpath=None):
        self._L = lib(libpath)
        # keep padded copies alive: the oracle aliases the images like the reference's mmap
        self._img = _padded(index_image, 64)
        self._wand = _padded(wand_image, 8) if wand_image is not None else None
        self._h = self._L.oracle_index_open(_codec(kind), _p(self._img), len(index_image),
                                            _p(self._wand) if self._wand is not None else None,
                                            len(wand_image) if wand_image is not None else 0)
        if not self._h:
            raise RuntimeError("oracle_index_open failed")

    def size(self):
        return self._L.oracle_index_size(self._h)

    def num_docs(self):
        return self._L.oracle_index_num_docs(self._h)

    def list_offset(self, term):
        return self._L.oracle_list_offset(self._h, term)

    def list_size(self, term):
        return self._L.oracle_list_size(self._h, term)

    def enumerate(self, term):
        n = self.list_size(term)
        d = np.zeros(n, dtype=np.uint32)
        f = np.zeros(n, dtype=np.uint32)
        r = self._L.oracle_list_enumerate(self._h, term, _p(d), _p(f), n)
        if r != n:
            raise RuntimeError("oracle_list_enumerate failed (%d)" % r)
        return d, f

    def next_geq(self, term, probes):
        pr = np.ascontiguousarray(probes, dtype=np.uint32)
        d = np.zeros(len(pr), dtype=np.uint32)
        f = np.zeros(len(pr), dtype=np.uint32)
        if self._L.oracle_list_next_geq(self._h, term, _p(pr), len(pr), _p(d), _p(f)):
            raise RuntimeError("oracle_list_next_geq failed")
        return d, f

    def move(self, term, positions):
        ps = np.ascontiguousarray(positions, dtype=np.uint32)
        d = np.zeros(len(ps), dtype=np.uint32)
        f = np.zeros(len(ps), dtype=np.uint32)
        if self._L.oracle_list_move(self._h, term, _p(ps), len(ps), _p(d), _p(f)):
            raise RuntimeError("oracle_list_move failed")
        return d, f

    def query(self, op, terms, k=10, want_matches=False, profile=False):
        t = np.ascontiguousarray(terms if len(terms) else [0], dtype=np.uint32)
        topk = np.full(k, -np.inf, dtype=np.float32)
        tl = C.c_uint32()
        cap = 0
        m = None
        if want_matches:
            cap = int(self.num_docs())
            m = np.zeros(cap, dtype=np.uint32)
        fs = C.c_uint64()
        prof = Profile()
        r = self._L.oracle_query(self._h, _op(op), k, _p(t), len(terms), _p(topk), C.byref(tl), _p(m) if m is not None else None,
                                 cap, C.byref(fs), C.byref(prof) if profile else None)
        if r < 0:
            raise RuntimeError("oracle_query failed (%d)" % r)
        out = {"count": r, "topk": topk[:tl.value].copy(), "freq_sum": fs.value}
        if want_matches:
            out["matches"] = m[:r].copy()
        if profile:
            out["profile"] = prof.as_dict()
        return out

    def query_batch(self, op, queries, k=10, profile=False):
        terms, offs = _flatten(queries)
        nq = len(queries)
        count = np.zeros(max(nq, 1), dtype=np.uint64)
        topk = np.full((max(nq, 1), k), -np.inf, dtype=np.float32)
        tlen = np.zeros(max(nq, 1), dtype=np.uint32)
        fsum = np.zeros(max(nq, 1), dtype=np.uint64)
        prof = Profile()
        rc = self._L.oracle_query_batch(self._h, _op(op), k, _p(terms), _p(offs), nq, _p(count), _p(topk), _p(tlen), _p(fsum),
                                        C.byref(prof) if profile else None)
        if rc:
            raise RuntimeError("oracle_query_batch failed (%d)" % rc)
        return count[:nq], topk[:nq], tlen[:nq], fsum[:nq], (prof.as_dict() if profile else None)

    def query_batch_mt(self, op, queries, k=10, threads=None, match_hash=False):
        """query_batch on `threads` host threads (default: the CPUs this process may use); no profile. With match_hash
        (and / and_freq) also returns, per query, sum_i doc_i * (2 i + 1) mod 2^64 over its doc-id list."""
        import os
        if threads is None:
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                threads = os.cpu_count() or 1
        terms, offs = _flatten(queries)
        nq = len(queries)
        count = np.zeros(max(nq, 1), dtype=np.uint64)
        topk = np.full((max(nq, 1), k), -np.inf, dtype=np.float32)
        tlen = np.zeros(max(nq, 1), dtype=np.uint32)
        fsum = np.zeros(max(nq, 1), dtype=np.uint64)
        mh = np.zeros(max(nq, 1), dtype=np.uint64) if match_hash else None
        rc = self._L.oracle_query_batch_mt(self._h, _op(op), k, _p(terms), _p(offs), nq, int(max(1, threads)), _p(count), _p(topk), _p(tlen),
                                           _p(fsum), _p(mh) if mh is not None else None)
        if rc:
            raise RuntimeError("oracle_query_batch_mt failed (%d)" % rc)
        return count[:nq], topk[:nq], tlen[:nq], fsum[:nq], (mh[:nq] if mh is not None else None)

    def perftest(self, op, queries, k=10, runs=2):
        """op_perftest (queries.cpp:13-62): returns dict(avg,q50,q90,q95 in microseconds, seconds)."""
        terms, offs = _flatten(queries)
        st = np.zeros(5, dtype=np.float64)
        rc = self._L.oracle_perftest(self._h, _op(op), k, _p(terms), _p(offs), len(queries), runs, _p(st))
        if rc:
            raise RuntimeError("oracle_perftest failed (%d)" % rc)
        return {"avg": st[0], "q50": st[1], "q90": st[2], "q95": st[3], "seconds": st[4]}

    def close(self):
        if self._h:
            self._L.oracle_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
