"""Static check of the generated gfx950 code: registers that an inline-asm load still has in flight are never touched by a
compiler-generated instruction before the hand-written wait (tools/check_inflight_moves.py; the failure mode of kmeans_screen_kernel in round 3:
loop-carried asm loads, a v_mov at the back-edge in front of the s_waitcnt, wrong labels in 1 run of the GPU suite in 6)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "u2seg_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the sources with inline-asm loads into registers (global_load_* / ds_read_*)
SOURCES = ["kmeans", "conv_igemm", "wgrad_halo", "conv_tile", "conv_halo", "conv_stream", "wgrad_stream"]


def test_checker_flags_a_move_of_an_in_flight_register(tmp_path):
    import check_inflight_moves as chk

    bad = tmp_path / "bad.s"
    bad.write_text("kernel_a:\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n"
                   "\tv_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[8:9], v[20:23]\n\tv_mov_b64_e32 v[30:31], v[12:13]\n"
                   "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text("kernel_b:\n\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v[2:3], off\n\t;;#ASMEND\n"
                    "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_mov_b64_e32 v[30:31], v[12:13]\n\ts_endpgm\n")
    assert chk.scan(str(bad)) == 1
    assert chk.scan(str(good)) == 0


@pytest.mark.skipif(shutil.which(HIPCC) is None, reason="hipcc not available")
def test_no_compiler_moves_of_in_flight_asm_loads(tmp_path):
    import check_inflight_moves as chk

    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I.", "-I../../include",
             "-Wno-unused-result", "-S", "--cuda-device-only"]  # build.sh's flags
    procs = [(s, subprocess.Popen([HIPCC] + flags + [s + ".hip", "-o", str(tmp_path / (s + ".s"))], cwd=CSRC,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)) for s in SOURCES]
    for s, p in procs:
        assert p.wait(timeout=600) == 0, "hipcc -S failed for " + s
    for s in SOURCES:
        assert chk.scan(str(tmp_path / (s + ".s"))) == 0, "compiler-generated move of an in-flight register in " + s + ".hip"
