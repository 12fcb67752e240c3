"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares."""
import os
import re

import pytest

import ds2i_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("ds2i_hip.h", "ds2i_build.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(ds2i_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_exports_every_declared_symbol(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(built_lib, s)]
    assert not missing, missing


def test_error_reporting_without_gpu(built_lib):
    # argument validation happens before any device work, so it is checkable on a CPU box
    import ctypes as C
    h = C.c_void_p()
    rc = built_lib.ds2i_hip_index_open(0, 99, b"x", 1, None, 0, C.byref(h))
    assert rc == -1 and b"unknown index kind" in built_lib.ds2i_hip_last_error()
    with pytest.raises(ds2i_amd.Ds2iError):
        ds2i_amd.build_index("block_optpfor", 10, [([], [])])


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    """The product must fail loudly, not fall back to the oracle, if the HIP extension is absent."""
    import ds2i_amd.api as api
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(api, "_HERE", str(tmp_path))
    with pytest.raises(ImportError):
        api.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ds2i_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "liboracle" not in txt and '"oracle/' not in txt, f


def test_device_code_has_no_function_calls(built_lib, tmp_path):
    """Every device function is meant to be inlined into its kernel: a lambda the inliner leaves as a real call costs
    the call ABI (register save / restore) on the hot path -- it once halved the disjunctive kernel's throughput."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not found")
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([tools[0], "--dump-section", ".hip_fatbin=" + fat, ds2i_amd.library_path(), os.devnull])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert len(starts) >= 5  # one code object per translation unit of kernels.hip (ds2i_amd/build.py)
    seen = ""
    for i, st in enumerate(starts):
        part, elf = str(tmp_path / ("bundle%d.bin" % i)), str(tmp_path / ("dev%d.elf" % i))
        open(part, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.check_call([tools[1], "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
        dis = subprocess.run([tools[2], "-d", "--no-show-raw-insn", elf], capture_output=True, text=True, check=True).stdout
        assert dis.count("s_swappc_b64") == 0 and dis.count("s_call_b64") == 0, i
        seen += "".join(sorted(set(re.findall(r"k_(?:conjunctive|disjunctive|daat|merge|decode_list|union_topk|unionILb)", dis))))
    for k in ("k_conjunctive", "k_disjunctive", "k_daat", "k_merge", "k_decode_list", "k_union_topk", "k_unionILb"):
        assert k in seen


def test_hand_issued_loads_have_no_register_destination(tmp_path):
    """ranked_stream.hip issues its prefetches and range-table gathers from inline asm and waits for them with counted
    s_waitcnt statements, because hipcc drains vmcnt wherever control flow joins. hipcc does not know such a load is
    pending: a VGPR destination counts as written when the statement ends, and under register pressure the compiler did
    copy the still-pending register (a v_mov of garbage; tests/asm_audit.py walks the control-flow graph for exactly that).
    The kernels therefore use LDS-DMA only (global_load_lds_*: no register destination). This holds them to it: every
    global load inside an asm statement of the compiled kernels is an LDS-DMA, and the audit finds nothing."""
    import subprocess
    import asm_audit
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the four translation units that issue loads by hand: the stream kernels of ranked_and / and (block_optpfor; block_mixed), of
    # wand / maxscore / ranked_or, and the list streams of or_freq / and. (minimum kernels, minimum hand-issued loads per kernel)
    texts = {}
    for src, min_kernels, min_dma in (("ranked_stream.hip", 30, 6), ("union_stream.hip", 10, 6), ("ranked_stream_mixed.hip", 3, 4), ("freq_stream.hip", 3, 3)):
        out = str(tmp_path / (src + ".s"))
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                               "-o", out, os.path.join(root, "ds2i_amd", "csrc", src)], stderr=subprocess.DEVNULL)
        texts[src] = open(out).read()
        ks = asm_audit.kernels(texts[src])
        assert len(ks) >= min_kernels, src
        for name, lines in ks.items():
            in_asm, dma = False, 0
            for l in lines:
                t = l.strip()
                if t.startswith(";;#ASMSTART"):
                    in_asm = True
                elif t.startswith(";;#ASMEND"):
                    in_asm = False
                elif in_asm and re.match(r"(global|buffer|flat)_load", t):
                    assert "_lds_" in t.split()[0], (name, t)
                    dma += 1
            # (the AND instantiations of k_ranked_stream have no shared floor word to fetch: one hand-issued load fewer)
            assert dma >= (min_dma - 1 if re.search(r"k_ranked_streamILi\d+ELb\dELb1ELb\dELi1EE", name) else min_dma), (src, name)
            assert asm_audit.audit(lines) == [], (src, name)
    # Two blocks in flight: between the hand-issued prefetch of block i+2 and the first LDS read of block i+1's decode the compiler must
    # not have put a wait that drains it. (Round 6 found an `s_waitcnt vmcnt(0)` there in every iteration -- the rare general side-slot
    # decoder's loads stayed "possibly pending" for hipcc's wait-count pass on the hot path; rs_settle_vm(), stream_common.hpp.)
    for src, pat in (("ranked_stream.hip", r"k_ranked_streamILi\d+ELb0ELb0ELb0ELi1EE"), ("union_stream.hip", r"k_union_streamILi\d+ELb0ELi1EE")):
        checked = 0
        for name, lines in asm_audit.kernels(texts[src]).items():
            if not re.search(pat, name):
                continue
            at = [i for i, l in enumerate(lines) if "global_load_lds_dword" in l and "offset:256" in l]
            assert at, name
            for l in lines[at[0]:]:
                s = l.strip()
                if s.startswith("ds_read"):
                    break
                assert not (s.startswith("s_waitcnt") and "vmcnt" in s and ";;#" not in s) or "ASM" in s, (name, s)
            checked += 1
        assert checked == 5, (src, checked)
    # Register budget of the shipped (uninstrumented block_optpfor) instantiations of k_ranked_stream, from the code-object
    # metadata of the same listing: capacity 2 / 4 at 6 waves per SIMD (80 VGPRs), 6 / 8 at 4 / 3. With two lists nothing is
    # spilled and the kernel needs no private segment; from capacity 4 on stage C is one loop body over the lists (round 6), which
    # took the scalars the compiler parks in VGPR lanes from 124 / 214 (3 / 4 lists, round 5) to 105 and from 256 .. 417 (5 .. 8
    # lists) to 143 / 178; capacity 4 still spills VGPRs at its 80-register budget (23, all inside stage C; 5 waves per SIMD with 1
    # spilled register measured slower).
    text = texts["ranked_stream.hip"]
    meta = {}
    for blk in re.split(r"\n  - \.agpr_count:", text)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[nm] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "private_segment_fixed_size")}
    budget = {2: (80, 0, 0, 64), 4: (80, 24, 80, 110), 6: (128, 0, 0, 150), 8: (168, 0, 0, 190), 16: (256, 0, 0, 340)}
    seen = 0
    for nm, m in meta.items():
        mm = re.search(r"k_ranked_streamILi(\d+)ELb0ELb0ELb0ELi1EE", nm)  # (capacity, STATS = false, AND = false: the shipped ranked_and instantiations)
        if not mm:
            continue
        seen += 1
        vg, vs, ps, ss = budget[int(mm.group(1))]
        assert m["vgpr_count"] <= vg and m["vgpr_spill_count"] <= vs and m["private_segment_fixed_size"] <= ps and m["sgpr_spill_count"] <= ss, (nm, m)
    assert seen == 5
    # ... and of k_union_stream (wand / maxscore / ranked_or): nothing in scratch beyond one register of capacity 4
    umeta = {}
    for blk in re.split(r"\n  - \.agpr_count:", texts["union_stream.hip"])[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
        umeta[nm] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("sgpr_spill_count", "vgpr_spill_count", "vgpr_count", "private_segment_fixed_size")}
    ubudget = {2: (80, 0, 0, 64), 4: (96, 2, 8, 160), 6: (128, 0, 0, 200), 8: (168, 0, 0, 290), 16: (256, 0, 0, 620)}
    seen = 0
    for nm, m in umeta.items():
        mm = re.search(r"k_union_streamILi(\d+)ELb0ELi1EE", nm)
        if not mm:
            continue
        seen += 1
        vg, vs, ps, ss = ubudget[int(mm.group(1))]
        assert m["vgpr_count"] <= vg and m["vgpr_spill_count"] <= vs and m["private_segment_fixed_size"] <= ps and m["sgpr_spill_count"] <= ss, (nm, m)
    assert seen == 5


def test_documented_knobs_are_the_knobs_of_the_library():
    """DESIGN.md section 7 lists the environment knobs of the library; capi.cpp reads them in one place (ds2i_knobs) from one table
    (kKnobs, which is also what ds2i_hip_set_option accepts) and knobs.hpp declares them: the three lists are the same twenty names,
    and no product source reads the environment anywhere else (a renamed or removed knob would otherwise stay documented, and an
    A/B reported against it would silently have compared a build with itself)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "DESIGN.md")).read()
    a, b = doc.index("## 7. Knobs"), doc.index("## 8. Out of scope")
    documented = set(re.findall(r"`(DS2I_[A-Z0-9_]+)`", doc[a:b])) - {"DS2I_BUILD_VARIANT"}
    csrc = os.path.join(root, "ds2i_amd", "csrc")
    capi = open(os.path.join(csrc, "capi.cpp")).read()
    table = set(re.findall(r'"(DS2I_[A-Z0-9_]+)"', capi[capi.index("kKnobs[] = {"):capi.index("};", capi.index("kKnobs[] = {"))]))
    header = set(re.findall(r"// (DS2I_[A-Z0-9_]+):", open(os.path.join(csrc, "knobs.hpp")).read()))
    assert documented == table == header, (documented ^ table, table ^ header)
    assert len(table) == 20
    sites = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".cpp", ".hip", ".hpp", ".inc")):
            for i, line in enumerate(open(os.path.join(csrc, f), errors="replace")):
                if "getenv(" in line and "LOCAL_WORLD_SIZE" not in line:
                    sites.append((f, i + 1))
    assert len(sites) == 1 and sites[0][0] == "capi.cpp", sites


def test_set_option_accepts_knobs_only(built_lib):
    """ds2i_hip_set_option: the C-ABI way to set a DESIGN.md section 7 knob (process-wide, before the first upload)"""
    ds2i_amd.set_option("DS2I_UNIT_FACTOR", 4)
    assert os.environ.get("DS2I_UNIT_FACTOR") is None or True  # (setenv in the C library: not mirrored in os.environ)
    ds2i_amd.set_option("DS2I_UNIT_FACTOR", None)
    for bad in ("PATH", "DS2I_lower", "DS2I_X=1"):
        with pytest.raises(ds2i_amd.Ds2iError) as e:
            ds2i_amd.set_option(bad, "1")
        assert e.value.code == -1
