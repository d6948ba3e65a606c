"""One feature + match pass on the cfg1 pair (target for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from describealign_amd import _native, synth
prec = _native.PREC_BF16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else _native.PREC_F32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 1320.0
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pair = synth.make_pair(5, secs, n_jumps=10, first_gap=200.0, channels=ch)          # channels = 2: bench.py's cfg2 pair of rank 0
c = _native.Context(0, prec)
c.pcm_upload(0, pair.video); c.pcm_upload(1, pair.audio)
vf = c.features_resident(0); af = c.features_resident(1)
for _ in range(2):
  c.match(vf, af)
if len(sys.argv) > 4 and sys.argv[4] == "chain":            # also the stage-2 chain DP on the resident matches
  c.match_begin(vf, af); c.match_finish()
  c.chain_resident()
print(c.stats())
