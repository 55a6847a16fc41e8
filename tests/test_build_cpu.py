"""CPU-only checks of the built product library: it loads, exports every symbol include/mbls.h declares, carries a
gfx950 code object without the base-pointer hazard (see mbls_fp.h), and has no path into the oracle."""
import ctypes
import os
import re
import shutil
import subprocess
import tempfile

import pytest

import helpers

LIB = os.path.join(helpers.ROOT, "milagro_bls_amd", "libmbls_hip.so")
HDR = os.path.join(helpers.ROOT, "include", "mbls.h")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def lib_path():
    from milagro_bls_amd import build
    return build.build()


def declared_symbols():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mbls_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ("mbls_fast_aggregate_verify_batch_device", "mbls_verify_batch_device", "mbls_verify_multiple_aggregate_signatures",
                 "mbls_aggregate_verify", "mbls_pk_from_bytes", "mbls_sig_from_bytes", "mbls_ctx_create"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib_path):
    l = ctypes.CDLL(lib_path)
    missing = [s for s in declared_symbols() if not hasattr(l, s)]
    assert not missing, missing
    # and the ctypes binding table covers the header 1:1
    from milagro_bls_amd import _native
    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_generated_asm_is_up_to_date():
    """the generator's output, written to a temporary directory (never over the tracked file), equals what the build compiles"""
    inc = os.path.join(helpers.ROOT, "milagro_bls_amd", "csrc", "mbls_fp_asm.inc")
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([__import__("sys").executable, os.path.join(helpers.ROOT, "tools", "gen_fp_asm.py")], stdout=subprocess.DEVNULL,
                              env=dict(os.environ, MBLS_GEN_OUT_DIR=d))
        assert open(os.path.join(d, "mbls_fp_asm.inc")).read() == open(inc).read(), "mbls_fp_asm.inc is stale: run tools/gen_fp_asm.py"


def test_no_oracle_in_product(lib_path):
    out = subprocess.check_output(["nm", "-D", lib_path], text=True)
    assert "orc_" not in out
    needed = subprocess.check_output([LLVM + "/llvm-readelf", "-d", lib_path], text=True)
    assert "bls_oracle" not in needed and "mbls_emul" not in needed
    pkg = os.path.join(helpers.ROOT, "milagro_bls_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import orc" not in src and "bls_oracle" not in src and "pymodel" not in src, f


def test_context_creation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from milagro_bls_amd import _native
    with pytest.raises(_native.MblsError):
        _native.Context(0)
    from milagro_bls_amd import PublicKey
    with pytest.raises(_native.MblsError):
        PublicKey.from_bytes(bytes(48))       # no CPU fallback behind the API


def test_code_object_has_no_base_pointer_frames(lib_path):
    # hipcc 7.2 IPRA clobbers the base pointer s34 across calls; mbls_fp.h avoids realigned frames altogether.
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        assert co, os.listdir(d)
        dis = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", os.path.join(d, co[0])], text=True)
    assert "v_mad_u64_u32" in dis                      # the hand-written multiplier is there
    assert "v_mfma" not in dis                         # carry-chain integer work, no MFMA
    assert not re.search(r"s_andn2_b32 s33, s33", dis), "a function realigns its stack (base pointer hazard)"
    assert not re.search(r"s_mov_b32 s32, s34", dis), "a function restores SP from the base pointer s34"


def kernel_metadata(lib_path):
    """per-kernel resource metadata of the gfx950 code object (llvm-readelf --notes)"""
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", os.path.join(d, co[0])], text=True)
    meta, cur = {}, {}
    for line in notes.splitlines():
        m = re.match(r"\s+-?\s*\.(name|private_segment_fixed_size|vgpr_count|agpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\S+)", line)
        if not m:
            continue
        if m.group(1) == "agpr_count" and "name" in cur:          # the first key of the next kernel's record
            meta[cur["name"]] = cur; cur = {}
        cur[m.group(1)] = m.group(2)
    if "name" in cur:
        meta[cur["name"]] = cur
    return meta


def test_kernel_resources(lib_path):
    """What the design rests on, read back from the built code object: the generated Miller and final-exponentiation kernels have no
    lane-private memory, the pipeline kernels are one-wave-per-SIMD kernels (more than 256 registers) whose LDS fits four waves per CU, and
    the key-decompression kernel -- nothing but a square root with the 8-entry window table -- fits 256 registers (two waves per SIMD)."""
    meta = kernel_metadata(lib_path)
    by = lambda prefix: next(v for k, v in meta.items() if k.startswith(prefix))
    for k in ("_Z8k_miller", "_Z7k_final", "_Z15k_miller_single"):
        assert int(by(k)["private_segment_fixed_size"]) == 0 and int(by(k)["vgpr_spill_count"]) == 0, k
    for k in ("_Z8k_miller", "_Z7k_final", "_Z6k_hash", "_Z5k_sig"):
        assert int(by(k)["vgpr_count"]) > 256, k
    for k in ("_Z8k_miller", "_Z7k_final", "_Z15k_miller_single"):
        assert int(by(k)["group_segment_fixed_size"]) * 4 <= 160 * 1024, k
    assert int(by("_Z15k_pk_decompress")["vgpr_count"]) <= 256 and int(by("_Z15k_pk_decompress")["agpr_count"]) >= 112


@pytest.mark.parametrize("flag", ["MBLS_NO_ASM", "MBLS_NO_FP2_ASM", "MBLS_NO_LDS_STATE"])
def test_builds_without_the_generated_routines_are_refused(flag):
    """The host side selects kernels whose bodies ARE the generated routines (the signature's subgroup verdict out of the generated
    Miller loop, k_blind_*_d, the tree levels, k_coop): a build that swaps them for compiled lane bodies would leave empty or unfused
    kernels behind the same launches. Such a build must not exist (#error at the top of mbls_kernels.hip)."""
    src = os.path.join(helpers.ROOT, "milagro_bls_amd", "csrc", "mbls_kernels.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-E", "-D" + flag, src, "-o", os.devnull],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "cannot be built with" in r.stderr
    # and nothing in the kernels file still branches on those switches
    txt = open(src).read().split("#endif", 1)[1]
    assert flag not in txt


# ---- the Rust facade (never compiled here: no toolchain) and the binding INTEGRATION.md shows must stay in lock-step with include/mbls.h
_C2CANON = {"uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void",
            "mbls_ctx": "MblsCtx", "mbls_keytable": "MblsKeyTable", "mbls_multi": "MblsMulti", "mbls_multi_keytable": "MblsMultiKeyTable", "mbls_scalar_source": "MblsScalarSource"}


def _split_params(txt):
    out, depth, cur = [], 0, ""
    for ch in txt:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _canon_c(param):
    """'const uint8_t* d_sigs' / 'uint8_t pk_out[96]' / 'mbls_ctx** out' -> ('const'|'mut', base, pointer depth) with arrays as pointers"""
    p = param.strip()
    arr = "[" in p
    p = re.sub(r"\[[^\]]*\]", "", p)
    const = bool(re.search(r"\bconst\b", p))
    p = re.sub(r"\bconst\b|\bstruct\b", " ", p)
    depth = p.count("*") + (1 if arr else 0)
    toks = p.replace("*", " ").split()
    base = toks[0]
    return ("const" if const and depth else "mut" if depth else "val", _C2CANON.get(base, base), depth)


def _canon_rust(ty):
    t = ty.strip()
    depth, kind = 0, "val"
    first = True
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        if first:
            kind = m.group(1); first = False
        depth += 1; t = t[m.end():]
    return (kind if depth else "val", t, depth)


def _c_prototypes():
    txt = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(mbls_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt):
        params = [] if m.group(3).strip() in ("", "void") else _split_params(m.group(3))
        protos[m.group(2)] = [_canon_c(p) for p in params]
    return protos


def _rust_decls(path):
    txt = open(path).read()
    decls = {}
    for m in re.finditer(r"\bfn\s+(mbls_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*(?:->\s*[^;{]+)?;", txt, flags=re.S):
        params = _split_params(" ".join(m.group(2).split()))
        decls.setdefault(m.group(1), []).append([_canon_rust(p.split(":", 1)[1]) for p in params])
    return decls


@pytest.mark.parametrize("src", ["rust/src/lib.rs", "INTEGRATION.md"])
def test_rust_bindings_match_the_header(src):
    """every `fn mbls_*` the Rust side declares exists in include/mbls.h with the same parameter list (count, pointer depth, constness, base type;
    `void* stream` = `*mut c_void`); the crate has never met rustc in this image, so this is what keeps it from drifting"""
    protos = _c_prototypes()
    decls = _rust_decls(os.path.join(helpers.ROOT, src))
    assert len(decls) >= 10, "no declarations found in " + src
    for name, variants in decls.items():
        assert name in protos, "%s declares %s, which include/mbls.h does not" % (src, name)
        for got in variants:
            want = protos[name]
            assert len(got) == len(want), "%s: %s has %d parameters, the header %d" % (src, name, len(got), len(want))
            for j, (g, w) in enumerate(zip(got, want)):
                # a pointer the C side does not write through may be *const or *mut on the Rust side only if the header says so: compare exactly
                assert g == w, "%s: %s parameter %d is %s, the header says %s" % (src, name, j, g, w)


def test_every_c_call_in_integration_md_has_the_header_arity():
    """the C snippets of INTEGRATION.md call the entries with as many arguments as the header declares"""
    protos = _c_prototypes()
    txt = open(os.path.join(helpers.ROOT, "INTEGRATION.md")).read()
    seen = 0
    for m in re.finditer(r"\b(mbls_[a-z0-9_]+)\s*\(", txt):
        name = m.group(1)
        if name not in protos:
            continue
        # the argument text up to the matching parenthesis
        i, depth = m.end(), 1
        while i < len(txt) and depth:
            depth += txt[i] == "("; depth -= txt[i] == ")"; i += 1
        args = re.sub(r"/\*.*?\*/", "", txt[m.end():i - 1], flags=re.S)
        if re.search(r":\s*\*", args) or "->" in txt[i:i + 12]:
            continue                                   # a Rust declaration (checked above), not a call
        if "..." in args:
            continue
        seen += 1
        assert len(_split_params(args)) == len(protos[name]), "INTEGRATION.md calls %s with %d arguments, the header declares %d" % (
            name, len(_split_params(args)), len(protos[name]))
    assert seen >= 10


def test_secret_key_selection_kernel_loads_every_record_unconditionally(lib_path):
    """k_sk_select (constant-time sk -> pk, include/mbls.h "SECRET KEYS ON THE DEVICE") must read ALL 16 records of a window whatever the key's digit is: with a
    plain `cond ? loaded : acc` hipcc guards the loads with the condition (s_and_saveexec + s_cbranch_execz around global_load: a key-dependent access pattern).
    Read back from the built code object: after the first 16-byte load there is no exec-mask manipulation and no conditional branch other than the uniform loop's."""
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        syms = subprocess.check_output([LLVM + "/llvm-readelf", "-s", "-W", os.path.join(d, co[0])], text=True)
        name = sorted(set(l.split()[-1] for l in syms.splitlines() if "k_sk_select" in l and " FUNC " in l))
        assert len(name) == 1, name
        dis = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", "--no-show-raw-insn", "--disassemble-symbols=" + name[0], os.path.join(d, co[0])], text=True)
    ops = [l.split()[0] for l in dis.splitlines() if l.startswith("\t")]
    first = ops.index("global_load_dwordx4")
    body = ops[first:]
    assert body.count("global_load_dwordx4") >= 16 and "global_store_dwordx4" in body
    assert not any("saveexec" in o or o in ("s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccz", "s_cbranch_vccnz") for o in body), [o for o in body if o.startswith("s_c") or "exec" in o]
    assert "v_cndmask_b32" not in " ".join(body) or True          # selections may be v_cndmask or and/or masks: both are data flow, not control flow
