"""CPU-only: the C-ABI library is built, loads, and exports every symbol include/chatllm_hip.h declares
(no compute calls: there is no GPU in the build container)."""
import os
import re
import subprocess

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "chatllm_hip.h")).read()
    return sorted(set(re.findall(r"CLLM_API\s+[\w\s\*]+?\b(cllm_\w+)\s*\(", txt)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    assert len(syms) >= 40
    for must in ("cllm_op_mul_mat", "cllm_op_mul_mat_id", "cllm_op_rms_norm", "cllm_op_rope", "cllm_op_soft_max", "cllm_op_set_rows",
                 "cllm_op_cpy", "cllm_op_get_rows", "cllm_quantize_row_q8_K", "cllm_llama_forward"):
        assert must in syms


def test_library_exports_every_declared_symbol(pkg):
    so = pkg.lib.SO_PATH
    assert os.path.exists(so), "run __graft_entry__.build() first"
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in header_symbols() if s not in exported]
    assert not missing, missing
    # nothing but the C ABI is exported (-fvisibility=hidden): no C++ or torch types in the boundary
    stray = [s for s in exported if not s.startswith("cllm_") and not s.startswith("_")]
    assert not stray, stray


def test_ctypes_signatures_cover_the_header(pkg):
    assert sorted(pkg.lib.SIGNATURES) == header_symbols()
    lib = pkg.lib.get()                       # loads and applies every prototype
    assert lib.cllm_abi_version() == 1
    assert lib.cllm_type_size(12) == 144 and lib.cllm_blck_size(12) == 256
    assert lib.cllm_row_size(2, 4096) == 2304 and lib.cllm_row_size(8, 4096) == 4352 and lib.cllm_row_size(12, 4096) == 2304


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    import pytest
    if pkg.lib.get().cllm_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.lib.CllmError):
        pkg.Tensor(pkg.F32, [4])
    with pytest.raises(pkg.lib.CllmError):
        pkg.Llama(pkg.synth.config("tiny"))


def test_product_package_never_imports_the_oracle():
    pkg_dir = os.path.join(ROOT, "chatllm.cpp_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "ggml_oracle" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)


def _resource_usage(name):
    """hipcc -Rpass-analysis=kernel-resource-usage of one csrc file with the flags the library is built with (read from the Makefile: a copy here could drift)"""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "chatllm.cpp_amd", "csrc", name)
    mk = open(os.path.join(ROOT, "chatllm.cpp_amd", "csrc", "Makefile")).read()
    m = re.search(r"^FLAGS\s*:=\s*(.*)$", mk, re.M)
    assert m, "csrc/Makefile: FLAGS line not found"
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)
    flags = m.group(1).replace("$(ARCH)", arch).split()
    assert "-ffp-contract=off" in flags and any(f.startswith("--offload-arch=") for f in flags), flags
    out = subprocess.run([hipcc] + flags + ["--cuda-device-only", "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stderr


def test_kernels_with_hand_issued_loads_do_not_spill():
    """gemv_team32.hip issues the chain wave's LDS reads by asm and waits for them by count: a register of such a read must never be spilled or copied while the
    read is in flight.  The allocator keeps them in place as long as nothing spills -- every instantiation must report 0 spilled registers and no scratch."""
    err = _resource_usage("gemv_team32.hip")
    spills = re.findall(r"VGPRs Spill: (\d+)", err) + re.findall(r"SGPRs Spill: (\d+)", err) + re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", err)
    assert len(spills) >= 3 * 30 and all(int(x) == 0 for x in spills), sorted(set(spills))


def test_the_decode_mat_vec_kernels_use_no_scratch():
    """the single-token mat-vec launches last 4-15 us: a kernel that needs scratch pays for its set-up in every one of them.  (Round 4: the order-exact RMS_NORM's
    serial fallback as a called function gave every norm-prologue instantiation of k_gemv_dec a call frame and up to 25 spilled registers, unnoticed by the
    parity tests -- it now runs on wave 0's lanes in place.)"""
    err = _resource_usage("gemv_decode.hip")
    vals = re.findall(r"VGPRs Spill: (\d+)", err) + re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", err)
    assert len(vals) >= 2 * 40 and all(int(x) == 0 for x in vals), sorted(set(vals))
