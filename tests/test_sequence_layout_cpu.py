"""The reference's own LAYOUT tests for the Elias-Fano family, run over the bits the product writers emit
(host_index.hpp::ef_write, host_pef.hpp::rb_write / write_partition_table):

  test/test_compact_elias_fano.cpp:45-80        every pointers0 / pointers1 entry against a direct scan of the high bits,
                                                every element from its high + low bits (sampling 4 / 5, "high granularity")
  test/test_compact_ranked_bitvector.cpp:36-68  every pointers1 / rank1_samples entry against a scan of the bitmap (6 / 5)
  test/test_partitioned_sequence.cpp:13-111     switch_partition(): base / upper bound / elements of every partition,
                                                singletons, avg gaps 1.1 .. 10, short sequences in a large universe
  test/test_uniform_partitioned_sequence.cpp    the same inputs for fixed partitions
  test/test_generic_sequence.hpp:28-164         move / next / prev_value / next_geq of every sequence type

The device never reads the sampled pointers (it decodes whole chunks), so nothing else in the suite would notice a writer
that filled them wrongly -- and real ds2i, which navigates by them, could not read such a file. The scans below are
written from the field definitions (compact_elias_fano.hpp:18-42 `offsets`, compact_ranked_bitvector.hpp:15-33) and share
no code with the writers; the enumerator checks run the oracle's readers (oracle_pef.hpp), which do navigate by pointers.
"""
import numpy as np
import pytest

import ds2i_amd as d
import oracle as o
from ds2i_amd.api import SEQUENCE_KINDS, write_sequence

FINE = (4, 5, 6, 5, 7)      # ef_log_sampling0/1, rb_log_rank1_sampling, rb_log_sampling1, log_partition_size
DEFAULT = (9, 8, 9, 8, 7)   # global_parameters.hpp:5-31


def random_sequence(universe, n, strict=True, seed=42):
    """test_generic_sequence.hpp:7-26 (the reference draws with rand(); any fixed stream will do)"""
    rng = np.random.default_rng(seed)
    u = universe - n if strict else universe
    seq = np.sort(rng.integers(0, u, n, dtype=np.uint64)) if u > 0 else np.zeros(n, dtype=np.uint64)
    if strict:
        seq = seq + np.arange(n, dtype=np.uint64)
    return seq


def ceil_log2(x):
    return 0 if x <= 1 else int(x - 1).bit_length()


def unpack(words, nbits):
    """bit i of the string = (words[i / 64] >> (i % 64)) & 1 (succinct::bit_vector)"""
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:nbits]


def get_field_array(bits, offset, count, width):
    """`count` consecutive `width`-bit little-endian fields starting at bit `offset`"""
    if count == 0 or width == 0:
        return np.zeros(count, dtype=np.uint64)
    f = bits[offset:offset + count * width].reshape(count, width).astype(np.uint64)
    return (f << np.arange(width, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)


def get_bits(bits, pos, width):
    return int(get_field_array(bits, pos, 1, width)[0])


class EfOffsets:  # compact_elias_fano.hpp:18-42
    def __init__(self, base, universe, n, params):
        self.ls0, self.ls1 = params[0], params[1]
        self.lower_bits = (universe // n).bit_length() - 1 if universe > n else 0
        self.higher_bits_length = n + (universe >> self.lower_bits) + 2
        self.pointer_size = ceil_log2(self.higher_bits_length)
        self.pointers0 = (self.higher_bits_length - n) >> self.ls0 if self.ls0 < 64 else 0
        self.pointers1 = n >> self.ls1
        self.pointers0_offset = base
        self.pointers1_offset = self.pointers0_offset + self.pointers0 * self.pointer_size
        self.higher_bits_offset = self.pointers1_offset + self.pointers1 * self.pointer_size
        self.lower_bits_offset = self.higher_bits_offset + self.higher_bits_length
        self.end = self.lower_bits_offset + n * self.lower_bits


class RbOffsets:  # compact_ranked_bitvector.hpp:15-33
    def __init__(self, base, universe, n, params):
        self.lr, self.ls1 = params[2], params[3]
        self.rank1_sample_size = ceil_log2(n + 1)
        self.pointer_size = ceil_log2(universe)
        self.rank1_samples = universe >> self.lr if self.lr < 64 else 0
        self.pointers1 = n >> self.ls1
        self.rank1_samples_offset = base
        self.pointers1_offset = self.rank1_samples_offset + self.rank1_samples * self.rank1_sample_size
        self.bits_offset = self.pointers1_offset + self.pointers1 * self.pointer_size
        self.end = self.bits_offset + universe


def check_elias_fano_image(bits, base, universe, seq, params):
    """test_compact_elias_fano.cpp:45-80 for the EF image at bit `base`; returns its end offset"""
    n = len(seq)
    of = EfOffsets(base, universe, n, params)
    assert of.end <= len(bits)
    high = bits[of.higher_bits_offset:of.higher_bits_offset + of.higher_bits_length]
    ones = np.flatnonzero(high)      # pos of the rank-th one
    zeros = np.flatnonzero(high == 0)
    assert len(ones) == n
    # every element from its high and low bits: ((pos - rank - 1) << l) | low[rank]
    low = get_field_array(bits, of.lower_bits_offset, n, of.lower_bits)
    rank = np.arange(n, dtype=np.uint64)
    assert np.array_equal(((ones.astype(np.uint64) - rank - 1) << np.uint64(of.lower_bits)) | low, np.asarray(seq, np.uint64))
    # pointers1[k-1] = position of the one with rank k << ls1 (k >= 1)
    p1 = get_field_array(bits, of.pointers1_offset, of.pointers1, of.pointer_size)
    sampled1 = np.arange(1, of.pointers1 + 1) << of.ls1
    assert np.all(sampled1 < n) or of.pointers1 == 0 or sampled1[-1] <= n
    valid = sampled1 < n
    assert np.array_equal(p1[valid], ones[sampled1[valid]].astype(np.uint64))
    assert np.all(p1[~valid] == 0)  # (n a multiple of the sampling: the last slot has no element and stays zero)
    # pointers0[k-1] = position of the zero with rank0 k << ls0 (k >= 1), rank0 = zeros before it
    p0 = get_field_array(bits, of.pointers0_offset, of.pointers0, of.pointer_size)
    sampled0 = np.arange(1, of.pointers0 + 1) << of.ls0 if of.pointers0 else np.zeros(0, np.int64)
    valid = sampled0 < len(zeros)
    assert np.array_equal(p0[valid], zeros[sampled0[valid]].astype(np.uint64))
    assert np.all(p0[~valid] == 0)
    return of.end


def check_ranked_bitvector_image(bits, base, universe, seq, params):
    """test_compact_ranked_bitvector.cpp:36-68"""
    n = len(seq)
    of = RbOffsets(base, universe, n, params)
    assert of.end <= len(bits)
    bm = bits[of.bits_offset:of.bits_offset + universe]
    ones = np.flatnonzero(bm)
    assert np.array_equal(ones.astype(np.uint64), np.asarray(seq, np.uint64))
    p1 = get_field_array(bits, of.pointers1_offset, of.pointers1, of.pointer_size)
    sampled1 = np.arange(1, of.pointers1 + 1) << of.ls1
    valid = sampled1 < n
    assert np.array_equal(p1[valid], ones[sampled1[valid]].astype(np.uint64))
    # rank1_samples[k-1] = ones before position k << lr (k >= 1, position < universe)
    rs = get_field_array(bits, of.rank1_samples_offset, of.rank1_samples, of.rank1_sample_size)
    pos = np.arange(1, of.rank1_samples + 1) << of.lr if of.rank1_samples else np.zeros(0, np.int64)
    valid = pos < universe
    assert np.array_equal(rs[valid], np.searchsorted(ones, pos[valid]).astype(np.uint64))
    return of.end


def oracle_ok(kind, words, nbits, universe, seq, params):
    rc = o.sequence_selftest(SEQUENCE_KINDS.index(kind), words, nbits, universe, seq, params)
    assert rc == 0, "%s: requirement %d of the restated reference test failed" % (kind, rc)


@pytest.mark.parametrize("params", [FINE, DEFAULT], ids=["sampling_4_5", "default_sampling"])
def test_compact_elias_fano_construction(built_lib, params):
    n = 100000
    universe = n * 1024
    seq = random_sequence(universe, n)
    words, nbits = write_sequence("elias_fano", seq, universe, params)
    bits = unpack(words, nbits)
    assert check_elias_fano_image(bits, 0, universe, seq, params) == nbits
    oracle_ok("elias_fano", words, nbits, universe, seq, params)   # compact_elias_fano_enumerator


def test_compact_elias_fano_singleton_and_weakly_monotone(built_lib):
    for universe, v in ((1, 0), (2, 1)):   # test_compact_elias_fano.cpp:34-43
        seq = np.array([v], np.uint64)
        words, nbits = write_sequence("elias_fano", seq, universe, FINE)
        check_elias_fano_image(unpack(words, nbits), 0, universe, seq, FINE)
        oracle_ok("elias_fano", words, nbits, universe, seq, FINE)
    n = 100000
    universe = n * 3                        # 90-97: duplicates allowed
    seq = random_sequence(universe, n, strict=False)
    assert len(np.unique(seq)) < n
    words, nbits = write_sequence("elias_fano", seq, universe, FINE)
    check_elias_fano_image(unpack(words, nbits), 0, universe, seq, FINE)
    oracle_ok("elias_fano", words, nbits, universe, seq, FINE)


@pytest.mark.parametrize("params", [FINE, DEFAULT], ids=["sampling_6_5", "default_sampling"])
def test_compact_ranked_bitvector_construction(built_lib, params):
    n = 100000
    universe = n * 3
    seq = random_sequence(universe, n, strict=True)
    words, nbits = write_sequence("ranked_bitvector", seq, universe, params)
    assert check_ranked_bitvector_image(unpack(words, nbits), 0, universe, seq, params) == nbits
    oracle_ok("ranked_bitvector", words, nbits, universe, seq, params)
    for universe, v in ((1, 0), (2, 1)):   # test_compact_ranked_bitvector.cpp:70-80
        s1 = np.array([v], np.uint64)
        w1, nb1 = write_sequence("ranked_bitvector", s1, universe, params)
        check_ranked_bitvector_image(unpack(w1, nb1), 0, universe, s1, params)
        oracle_ok("ranked_bitvector", w1, nb1, universe, s1, params)


# ---- partitioned sequences: an independent reader of the header (partitioned_sequence.hpp:131-178)
class BitCursor:
    def __init__(self, bits, pos=0):
        self.bits, self.pos = bits, pos

    def take(self, width):
        v = get_bits(self.bits, self.pos, width)
        self.pos += width
        return v

    def gamma(self):  # integer_codes.hpp:21-31
        z = 0
        while not self.bits[self.pos]:
            self.pos += 1
            z += 1
        self.pos += 1
        return (self.take(z) | (1 << z)) - 1

    def delta(self):  # integer_codes.hpp:33-45
        l = self.gamma()
        return (self.take(l) | (1 << l)) - 1


def ef_values(bits, base, universe, n, params):
    of = EfOffsets(base, universe, n, params)
    ones = np.flatnonzero(bits[of.higher_bits_offset:of.higher_bits_offset + of.higher_bits_length]).astype(np.uint64)
    low = get_field_array(bits, of.lower_bits_offset, n, of.lower_bits)
    return ((ones - np.arange(n, dtype=np.uint64) - 1) << np.uint64(of.lower_bits)) | low, of.end


def base_sequence_values(bits, pos, universe, n, params, strict):
    """indexed_sequence.hpp:89-164 / strict_sequence.hpp:98-174: type bit, then EF / ranked bitvector / nothing. Checks the
    nested image's pointers too. Returns the n values."""
    sp = list(params)
    if strict:
        sp[0] = 63
        sp[2] = 63
    if universe == n:
        return np.arange(n, dtype=np.uint64)
    t = int(bits[pos])
    if t == 0:
        u = universe - n + 1 if strict else universe
        vals, _ = ef_values(bits, pos + 1, u, n, sp)
        check_elias_fano_image(bits, pos + 1, u, vals, sp)
        return vals + np.arange(n, dtype=np.uint64) if strict else vals
    of = RbOffsets(pos + 1, universe, n, sp)
    vals = np.flatnonzero(bits[of.bits_offset:of.bits_offset + universe]).astype(np.uint64)
    assert len(vals) == n
    check_ranked_bitvector_image(bits, pos + 1, universe, vals, sp)
    return vals


def check_partitioned_image(bits, universe, seq, params, strict, uniform):
    """the constructor + switch_partition() of partitioned_sequence.hpp:131-178, 300-330 (uniform: :120-160), written
    against the layout; asserts what partitioned_sequence_test::test_construction asserts"""
    n = len(seq)
    cur = BitCursor(bits)
    partitions = cur.gamma() + 1
    if partitions == 1:
        base = cur.take(ceil_log2(universe))
        ub = 0
        if n > 1:
            ud = cur.delta()
            ub = ud if ud else universe - base - 1
        assert base == seq[0] and base + ub == seq[-1]
        vals = base_sequence_values(bits, cur.pos, ub + 1, n, params, strict)
        assert np.array_equal(vals + np.uint64(base), seq)
        return 1
    endpoint_bits = cur.gamma()
    pos = cur.pos
    if uniform:
        psize = 1 << params[4]
        assert partitions == -(-n // psize)
        ends = np.minimum(np.arange(1, partitions + 1, dtype=np.uint64) * np.uint64(psize), np.uint64(n))
    else:
        sizes, nxt = ef_values(bits, pos, n, partitions - 1, params)
        check_elias_fano_image(bits, pos, n, sizes, params)
        pos = nxt
        ends = np.concatenate([sizes, [np.uint64(n)]])
        assert np.all(np.diff(np.concatenate([[np.uint64(0)], ends]).astype(np.int64)) > 0)
    ubs, nxt = ef_values(bits, pos, universe, partitions + 1, params)
    check_elias_fano_image(bits, pos, universe, ubs, params)
    pos = nxt
    endpoints = np.concatenate([[np.uint64(0)], get_field_array(bits, pos, partitions - 1, endpoint_bits)])
    sequences = pos + endpoint_bits * (partitions - 1)
    step = max(1, partitions // 300)   # every partition of short inputs, a sample of long ones
    for p in list(range(0, partitions, step)) + [partitions - 1]:
        begin = int(ends[p - 1]) if p else 0
        end = int(ends[p])
        cur_base = int(seq[begin - 1]) + 1 if p else int(seq[0])
        assert int(ubs[p]) + (1 if p else 0) == cur_base, p          # m_cur_base
        assert int(ubs[p + 1]) == int(seq[end - 1]), p               # m_cur_upper_bound
        vals = base_sequence_values(bits, sequences + int(endpoints[p]), int(ubs[p + 1]) - cur_base + 1, end - begin, params, strict)
        assert np.array_equal(vals + np.uint64(cur_base), seq[begin:end]), p
    return partitions


def partition_inputs():
    yield 1, np.array([0], np.uint64)            # singletons (test_partitioned_sequence.cpp:86-95)
    yield 2, np.array([1], np.uint64)
    for g in (1.1, 1.9, 2.5, 3, 4, 5, 10):        # :97-104
        n = 10000
        universe = int(n * g)
        yield universe, random_sequence(universe, n, True, seed=int(g * 10))
    rng = np.random.default_rng(7)
    for i in range(1, 512, 41):                   # :106-114 short sequences, large universe
        universe = 100000
        gap = int(rng.integers(0, 50000))
        yield universe, random_sequence(universe - gap, i, True, seed=i) + np.uint64(gap)
    # clustered: dense runs (all-ones / bitmap partitions) between sparse stretches -> many partitions of all three types
    parts, at = [], 0
    for r in range(60):
        ln = int(rng.integers(50, 700))
        if r % 3 == 0:
            parts.append(at + np.arange(ln))
            at += ln + int(rng.integers(1, 2000))
        else:
            g = np.cumsum(rng.integers(1, 3 if r % 3 == 1 else 400, ln))
            parts.append(at + g)
            at += int(g[-1]) + int(rng.integers(1, 2000))
    yield at + 17, np.concatenate(parts).astype(np.uint64)


@pytest.mark.parametrize("kind", ["partitioned_indexed", "partitioned_strict", "uniform_indexed", "uniform_strict"])
def test_partitioned_sequence_construction(built_lib, kind):
    strict = kind.endswith("strict")
    uniform = kind.startswith("uniform")
    total_partitions = 0
    for params in (DEFAULT, FINE):
        for universe, seq in partition_inputs():
            words, nbits = write_sequence(kind, seq, universe, params)
            bits = unpack(words, nbits)
            total_partitions += check_partitioned_image(bits, universe, seq, params, strict, uniform)
            oracle_ok(kind, words, nbits, universe, seq, params)   # test_construction + test_sequence with the oracle's reader
    assert total_partitions > 200   # the multi-partition path was exercised


@pytest.mark.parametrize("kind", ["indexed", "strict"])
def test_indexed_and_strict_sequences(built_lib, kind):
    """test_indexed_sequence.cpp / test_strict_sequence.cpp shapes: sparse (EF), dense (bitmap), all ones"""
    strict = kind == "strict"
    n = 20000
    for universe in (n, int(n * 1.5), n * 3, n * 500):
        seq = random_sequence(universe, n, True, seed=universe % 97)
        for params in (DEFAULT, FINE):
            words, nbits = write_sequence(kind, seq, universe, params)
            vals = base_sequence_values(unpack(words, nbits), 0, universe, n, params, strict)
            assert np.array_equal(vals, seq)
            oracle_ok(kind, words, nbits, universe, seq, params)
