"""Diagnostic: how uneven is the work across agents?  PROFILE=1 build; per-agent busy ticks over an
epoch of 800 calls.  Static spread (some agents always slow) bounds any asynchronous scheme by the
slowest agent's serial chain; dynamic spread (per call) is what a barrier per call pays for."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AZD_LIB"] = os.path.join(ROOT, "azdopt_amd", "libazdopt_amd_prof.so")
sys.path.insert(0, ROOT)
import numpy as np

import azdopt_amd as az

B, calls = 4096, 800
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
roots = space.generate_roots(0, B)
async_step = len(sys.argv) > 1 and sys.argv[1] == "async"
opt = az.NablaOptimizer.par_new(space, roots, model, B, async_step=async_step)
tol = ([200, 50, 50], 25)
k = np.array([sum(bin(int(w)).count("1") for w in row) for row in roots[1]])
prev = opt.agent_counters()
chunks, waits = [], []
opt.set_timing(True)
for c in range(8):
    opt.par_roll_out_episodes(tol, n_calls=calls // 8)
    cur = opt.agent_counters()
    chunks.append((cur["TICKS_TOTAL"] - prev["TICKS_TOTAL"]).astype(np.float64) / 100 / (calls // 8))
    waits.append((cur["TICKS_WAIT"] - prev["TICKS_WAIT"]).astype(np.float64) / 100 / (calls // 8))
    prev = cur
tm = opt.timing()
print("kernel time per call: %.1f us" % (tm["rollout_ms"] * 1e3 / calls))
wait = np.stack(waits).mean(0)
cyc = np.stack(chunks).mean(0) + wait
print("wait us per call per agent: mean %.1f p99 %.1f max %.1f;  busy+wait: mean %.1f p50 %.1f p99 %.1f max %.1f" % (
    wait.mean(), np.percentile(wait, 99), wait.max(), cyc.mean(), np.percentile(cyc, 50), np.percentile(cyc, 99), cyc.max()))
wgc = cyc.reshape(-1, 16)
print("busy+wait per workgroup: mean of max %.1f, max of max %.1f, min of max %.1f" % (wgc.max(1).mean(), wgc.max(1).max(), wgc.max(1).min()))
per = np.stack(chunks)  # [8 chunks][B] mean busy us per call
tot = per.mean(0)
print("busy us per call, per agent over the epoch: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f  (max/mean %.2f)" % (
    tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), tot.max() / tot.mean()))
print("correlation of an agent's busy time between the first and the second half of the epoch: %.3f" % np.corrcoef(per[:4].mean(0), per[4:].mean(0))[0, 1])
print("correlation with the number of permitted actions k: %.3f" % np.corrcoef(tot, k)[0, 1])
for lo, hi in ((5, 20), (20, 40), (40, 60), (60, 77)):
    m = (k >= lo) & (k < hi)
    print("  k in [%d,%d): %4d agents, mean busy %.1f us" % (lo, hi, m.sum(), tot[m].mean()))
wg = tot.reshape(-1, 16)
print("per workgroup of 16: mean of max %.1f us, mean of mean %.1f us, max of max %.1f" % (wg.max(1).mean(), wg.mean(1).mean(), wg.max(1).max()))
print("by chunk of 100 calls (mean busy us):", " ".join("%.1f" % x for x in per.mean(1)))
