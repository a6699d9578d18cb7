"""CPU suite: the C-ABI library builds for gfx950, loads, exports every symbol include/mi355_interp.h
declares, and fails loudly (no CPU fallback) when no GPU is present.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

from armadillocudalinearinterpolation_amd import _build, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355_interp.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:mi_status|const char\*|int|void|size_t|mi_ctx\*|mi_edm\*)\s+(mi_[a-z0-9_]+)\s*\(", text, flags=re.M)
    assert len(names) > 25
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    path = _build.build_lib()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "declared in mi355_interp.h but not exported: %s" % missing
    assert lib.mi_abi_version() == 4


def test_python_binding_table_matches_header():
    assert sorted(_lib.SIGNATURES) == declared_functions()
    _lib.load()            # strict: raises if any symbol is absent


def test_code_object_is_gfx950_only():
    out = subprocess.run(["strings", "-a", _build.LIB_PATH], capture_output=True, text=True).stdout
    archs = set(re.findall(r"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", out))
    assert archs == {"gfx950"}, archs


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu suite")
    import armadillocudalinearinterpolation_amd as mi
    with pytest.raises(mi.MiError) as e:
        mi.Context(0)
    assert e.value.code == 5 and "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "armadillocudalinearinterpolation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert not re.search(r'#include\s*[<"][^>"]*oracle', src), f
                assert "liboracle" not in src, f


def test_public_header_is_plain_c99(tmp_path):
    """include/mi355_interp.h is what a cgo / JNI / ctypes binding includes: it must compile as C99 with -pedantic,
    and a C program must link against the library using only that header."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "c_abi.c"
    src.write_text('#include "mi355_interp.h"\n'
                   "/* every multi-GPU entry point is referenced, so the link fails if one is missing or mis-declared */\n"
                   "typedef void (*fn)(void);\n"
                   "static fn group_api[] = { (fn)mi_group_create, (fn)mi_group_destroy, (fn)mi_group_size,\n"
                   "  (fn)mi_group_ctx, (fn)mi_group_synchronize, (fn)mi_group_set_reduce, (fn)mi_group_grid1_create,\n"
                   "  (fn)mi_group_grid1_destroy, (fn)mi_group_interp1_f64_host, (fn)mi_group_interp1_f64_dev,\n"
                   "  (fn)mi_group_grid2_create, (fn)mi_group_grid2_destroy, (fn)mi_group_interp2_f64_host, (fn)mi_group_interp2_f64_dev,\n"
                   "  (fn)mi_group_edm_create, (fn)mi_group_edm_destroy, (fn)mi_group_edm_set_params,\n"
                   "  (fn)mi_group_edm_compute_f, (fn)mi_group_edm_shard, (fn)mi_group_edm_shard_bounds };\n"
                   "int main(void) { mi_edm_params p; size_t lo, hi; double z[3] = {0.3, 0.7, 1.4}, f[3], sc[MI_EDM_PARTIAL_LEN(3)] = {3, 3, 3, 2, 9, 9, 9};\n"
                   "  mi_edm_default_params(&p); mi_shard_bounds(10, 1, 3, &lo, &hi);      /* host-only calls really run */\n"
                   "  if (mi_edm_residual_from_sums(&p, z, sc, f) != MI_OK) return 2;\n"
                   "  return (p.n_spikes == 3u && p.mean_quirk == 1 && lo == 4 && hi == 7 && group_api[0] &&\n"
                   "          mi_abi_version() == MI355_INTERP_ABI_VERSION) ? 0 : 1; }\n")
    exe = tmp_path / "c_abi"
    lib_dir = os.path.join(root, "armadillocudalinearinterpolation_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
           "-o", str(exe), "-L", lib_dir, "-lmi355interp", "-Wl,-rpath," + lib_dir]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)


def test_shard_bounds_host_arithmetic():
    """mi_shard_bounds is host-only (no GPU needed): contiguous, balanced, covering; equals sharding.shard_bounds."""
    from armadillocudalinearinterpolation_amd import sharding
    L = _lib.load()
    lo, hi = ctypes.c_size_t(0), ctypes.c_size_t(0)
    for n in (0, 1, 7, 8, 100, 10**8 + 3, 2**33 + 5):
        for world in (1, 2, 3, 8, 64):
            b = []
            for r in range(world):
                L.mi_shard_bounds(n, r, world, ctypes.byref(lo), ctypes.byref(hi))
                b.append((lo.value, hi.value))
            assert b == [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in b]
            assert max(sizes) - min(sizes) <= 1


def test_no_product_kernel_uses_scratch(tmp_path):
    """Every gfx950 kernel of libmi355interp.so runs out of registers and LDS alone: `.private_segment_fixed_size` is 0 in
    the code objects' metadata.  (A register spill would be a performance bug, and scratch brings the runtime's dynamic
    scratch management into a process that otherwise never needs it.)  Read with the LLVM tools that ship with ROCm."""
    import glob
    import re
    import shutil
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(os.path.join(llvm, "llvm-objdump")) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("ROCm's llvm-objdump / llvm-readelf not found")
    from armadillocudalinearinterpolation_amd import _build
    _build.build_lib()
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(_build.LIB_PATH, work / "lib.so")                       # (--offloading writes the bundles next to its input)
    out = subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", "lib.so"], cwd=work, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    bundles = glob.glob(str(work / "lib.so.*gfx950"))
    assert bundles, "no gfx950 code object in the library"
    kernels = 0
    for b in bundles:
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", b], capture_output=True, text=True).stdout
        sizes = re.findall(r"\.private_segment_fixed_size:\s*(\d+)", notes)
        names = re.findall(r"\.name:\s*(\S+)", notes)
        kernels += len(sizes)
        assert all(int(x) == 0 for x in sizes), [n for n in names if "evolve" in n or "interp" in n][:3]
    assert kernels >= 50          # (the evolve, interp1, interp2, restrict and probe kernels)
