"""Generates tests/golden/qmx_reference_blocks.json: 128-value blocks and the byte streams the REFERENCE QMX encoder
(/root/reference/qmx_codec.hpp, compiled as-is into oracle/_ref/libqmx_ref.so by oracle/Makefile) produces for them.
Run in the build container (needs /root/reference); the fixture travels, the reference does not.
Grid: the magnitudes of test_block_codecs.cpp:35-45 (1..24 bits) x four value shapes, seeded."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle as o

o.build()
R = o.ref_qmx()
assert R is not None, "oracle/_ref/libqmx_ref.so missing (needs /root/reference)"
rng = np.random.default_rng(12345)
cases = []
for mag in range(1, 25):
    for kind in range(4):
        if kind == 0:
            v = rng.integers(0, 1 << mag, size=128)
        elif kind == 1:
            v = np.where(rng.random(128) < 0.9, rng.integers(0, 4, 128), rng.integers(0, 1 << mag, 128))
        elif kind == 2:
            v = np.repeat(rng.integers(0, 1 << mag, 32), 4) >> rng.integers(0, 8, 128)
        else:
            v = np.where(rng.random(128) < 0.7, 1, rng.integers(0, 1 << max(1, mag // 2), 128))
        v = np.ascontiguousarray(v, dtype=np.uint32)
        buf = np.zeros(8192, dtype=np.uint8)
        ln = R.ref_qmx_encode(buf.ctypes.data, v.ctypes.data)
        out = np.zeros(640, dtype=np.uint32)
        R.ref_qmx_decode(out.ctypes.data, buf.ctypes.data, ln)
        assert np.array_equal(out[:128], v)
        cases.append({"mag": mag, "kind": kind, "values": v.tolist(), "hex": bytes(buf[:ln]).hex()})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qmx_reference_blocks.json")
json.dump({"source": "reference qmx_codec.hpp via oracle/_ref/libqmx_ref.so", "cases": cases}, open(path, "w"), separators=(",", ":"))
print(path, len(cases), os.path.getsize(path))
