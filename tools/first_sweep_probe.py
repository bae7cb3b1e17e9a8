"""Where the first C5 sweep of bench.py's process goes: the cca leg, then the loso leg, as bench.py runs them.
   python tools/first_sweep_probe.py [chain] [nocca] [profile]     chain: td_set_option cca_fused 0"""
import os, sys, json, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from telluride_decoding_amd import device
h = device.default_handle()
if 'chain' in sys.argv: h.set_option('cca_fused', 0)
eeg, env, offs = bench.make_workload(0)
if 'nocca' not in sys.argv: bench.cca_leg(h, device, eeg)
if 'profile' in sys.argv:
  pr = cProfile.Profile(); pr.enable()
leg = bench.loso_leg(eeg, env)
if 'profile' in sys.argv:
  pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
print(json.dumps({k: leg[k] for k in ('seconds', 'seconds_first_sweep', 'first_sweep_parts')}))
