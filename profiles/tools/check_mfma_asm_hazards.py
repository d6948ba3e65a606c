#!/usr/bin/env python3
"""Static check of the hand-placed (inline assembly) MFMAs of k_match_bf16_direct.

The compiler's hazard recogniser does not look inside inline assembly, so two rules are checked on
the generated ISA (hipcc -save-temps) instead:
  1. no VALU instruction writes a source VGPR of an MFMA within the two wait states in front of it
     ("VALU write VGPR -> MFMA read": 2 wait states on gfx90a and later);
  2. no VALU / LDS / VMEM instruction reads or overwrites an MFMA's destination registers before two
     further MFMAs have been issued behind it (the slot layout of bd_phase guarantees this; the check
     catches a compiler that moves an epilogue instruction up).
usage: check_mfma_asm_hazards.py <file.s> [kernel symbol]
"""
import re, sys

def regs(tok):
  m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
  if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
  m = re.fullmatch(r"v(\d+)", tok)
  if m: return {int(m.group(1))}
  return set()

def main():
  path = sys.argv[1]
  sym = sys.argv[2] if len(sys.argv) > 2 else "_ZN2da19k_match_bf16_directENS_9MatchArgsE"
  text = open(path).read()
  i = text.index(sym + ":"); j = text.index(".end_amdhsa_kernel", i)
  ins = []
  for line in text[i:j].split("\n"):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."): continue
    parts = line.replace(",", " ").split()
    ins.append((parts[0], parts[1:]))
  bad = 0
  n_mfma = 0
  for k, (op, args) in enumerate(ins):
    if not op.startswith("v_mfma"): continue
    n_mfma += 1
    dst = regs(args[0]); src = regs(args[2]) | (regs(args[3]) if len(args) > 3 else set())
    # rule 1
    ws, b = 0, k - 1
    while b >= 0 and ws < 2:
      o, a = ins[b]
      if o == "s_nop": ws += int(a[0]) + 1
      else:
        if o.startswith("v_") and not o.startswith("v_mfma") and a and (regs(a[0]) & (src - dst)):
          print("rule 1: %s %s writes a source of the MFMA %d instructions later" % (o, " ".join(a), k - b)); bad += 1
        ws += 1
      b -= 1
    # rule 2
    seen, f = 0, k + 1
    while f < len(ins) and seen < 2:
      o, a = ins[f]
      if o.startswith("v_mfma"):
        seen += 1
      elif o.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
        touched = set()
        for t in a: touched |= regs(t)
        if touched & dst:
          print("rule 2: %s %s touches the destination of an MFMA %d instructions earlier" % (o, " ".join(a), f - k)); bad += 1
      elif o.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        break                                     # the slot layout never branches inside the window; beyond it the distance only grows
      f += 1
  print("%d MFMAs checked, %d violations" % (n_mfma, bad))
  return 1 if bad else 0

if __name__ == "__main__":
  sys.exit(main())
