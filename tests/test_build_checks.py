"""Build-time guards (CPU: hipcc cross-compiles gfx950 without a GPU)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "describealign_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
KERNEL = "_ZN2da12k_match_bf16ENS_9MatchArgsE"
KERNEL_F32 = "_ZN2da11k_match_f32ENS_9MatchArgsE"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_bf16_gemm_inline_asm_mfma_rules(tmp_path):
  """k_match_bf16 and k_match_f32 place their MFMAs by inline assembly (accumulators in VGPRs, resident operand
  in AGPRs), so the compiler's hazard recogniser does not cover them.  Compile the file exactly as the Makefile
  does, keep the ISA, and check on it: the two hazard rules of profiles/tools/check_mfma_asm_hazards.py, no
  scratch, and no v_accvgpr_read / v_accvgpr_write in the kernels (the point of the register placement)."""
  flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M).group(1)
  flags = flags.replace("$(ARCH)", "gfx950").split()
  for name in ("dalign_match.hip", "dalign_common.h", "dalign_stretch.h"):
    shutil.copy(os.path.join(CSRC, name), tmp_path)
  cmd = [HIPCC] + flags + ["-save-temps=obj", "-c", "dalign_match.hip", "-o", "m.o"]
  r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-2000:]
  isa = os.path.join(tmp_path, "dalign_match-hip-amdgcn-amd-amdhsa-gfx950.s")
  text = open(isa).read()
  for kernel, min_mfma in ((KERNEL, 54), (KERNEL_F32, 252)):
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "check_mfma_asm_hazards.py"), isa, kernel],
                         capture_output=True, text=True)
    assert chk.returncode == 0 and "0 violations" in chk.stdout, chk.stdout[-3000:] + chk.stderr
    assert re.search(r"(\d+) MFMAs checked", chk.stdout) and int(re.search(r"(\d+) MFMAs checked", chk.stdout).group(1)) >= min_mfma
    body = text[text.index(kernel + ":"):]
    body = body[:body.index(".end_amdhsa_kernel") + 4000]
    assert re.search(r"; ScratchSize: 0\b", body), kernel + " spills"
    kernel_only = body[:body.index(".end_amdhsa_kernel")]
    assert "v_accvgpr_read" not in kernel_only and "v_accvgpr_write" not in kernel_only
    assert "s_set_gpr_idx" not in kernel_only, kernel + ": a fragment register is indexed at run time (a loop was not unrolled)"
