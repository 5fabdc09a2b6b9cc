"""Greedy per-layer search of the tile configuration / split-K factor of the UNet's convolutions with the two-image PAIR
pass (the frame's shape: two passes side by side on two streams) as the objective - isolated per-layer timings
(scripts/bench_conv.py) rank the tiles differently than the pair does (profiles/r04_experiments.md #18).

    python scripts/tune_pair_plan.py [H W [H2 W2]]        # on the GPU box; prints every evaluation

Every evaluation is its own process (PXT_CONV_PLAN is read once per process): ten pair passes back to back behind a spin
kernel (the host ahead of the device, as in the frame loop), median of five such runs."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
WORKER = r'''
import sys, torch
sys.path.insert(0, %r)
from pixtrack_amd.unet import UNet, make_synthetic_unet_weights
H, W, H2, W2 = (int(x) for x in sys.argv[1:5])
dev = torch.device("cuda:0")
net = UNet(make_synthetic_unet_weights(7), dev)
g = torch.Generator().manual_seed(1)
a = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8).to(dev)
b = (torch.rand(H2, W2, 3, generator=g) * 255).to(dev)
m = (torch.rand(H2, W2, generator=g) > 0.4).to(torch.uint8).to(dev)
for _ in range(4):
    net.forward_packed_batch([(a, None, False), (b, m, True)])
torch.cuda.synchronize()
# the frame loop's regime: the host AHEAD of the device.  A spin kernel parks the stream for ~4 ms while the host enqueues
# twelve pair passes back to back; events bracket the last ten on the device (a pass timed behind a synchronisation starts
# with the host still enqueueing and ranks the tiles differently: profiles/r04_experiments.md #18)
ts = []
for rep in range(5):
    torch.cuda._sleep(8_000_000)
    net.forward_packed_batch([(a, None, False), (b, m, True)])
    net.forward_packed_batch([(a, None, False), (b, m, True)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net.forward_packed_batch([(a, None, False), (b, m, True)])
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
ts.sort()
print("RESULT", ts[len(ts) // 2], ts[0])
''' % str(ROOT)


def plan_str(plan):
    return ";".join(f"{l}:{c}:{s}" for l, (c, s) in sorted(plan.items()) if c or s)


def evaluate(plan, shape, reps=1):
    best = None
    for _ in range(reps):
        env = dict(os.environ, PXT_CONV_PLAN=plan_str(plan))
        out = subprocess.run([sys.executable, "-c", WORKER, *map(str, shape)], env=env, capture_output=True, text=True, timeout=300)
        line = next((l for l in out.stdout.splitlines() if l.startswith("RESULT")), None)
        if line is None:
            return float("inf")  # (an invalid plan: the library refuses it)
        med = float(line.split()[1])
        best = med if best is None else min(best, med)
    return best


def main():
    a = [int(x) for x in sys.argv[1:]]
    shape = (a + [480, 640, 480, 640])[:2] + (a[2:4] if len(a) >= 4 else [480, 640]) if a else [480, 640, 480, 640]
    enc_cfgs = [1, 2, 4, 6, 11, 13, 14, 15, 17, 18, 19]
    cands = {l: [(c, 0) for c in enc_cfgs] for l in range(2, 13)}
    for l in (10, 11, 12):
        cands[l] = [(c, s) for c in (1, 2, 18, 19, 14) for s in (2, 4, 8)]
    cands[13] = [(2, s) for s in (0, 4, 7, 12)] + [(21, s) for s in (4, 8)] + [(17, s) for s in (4, 8)]
    cands[14] = [(c, s) for c in (2, 17, 21) for s in (0, 1, 2, 3)]
    cands[15] = [(c, s) for c in (2, 17, 21) for s in (0, 1, 2)]
    cands[16] = [(c, 0) for c in (6, 16, 20)]
    plan = {}
    base = evaluate(plan, shape, reps=2)
    print(f"shape {shape}  default plan: {base:.4f} ms", flush=True)
    for sweep in range(2):
        changed = False
        for l in sorted(cands):
            cur = evaluate(plan, shape, reps=2)
            best_c, best_t = None, cur
            for c in cands[l]:
                if plan.get(l) == c:
                    continue
                trial = dict(plan)
                trial[l] = c
                t = evaluate(trial, shape)
                print(f"  layer {l:2d} cfg {c[0]:2d} splits {c[1]}: {t:.4f} ms (current {cur:.4f})", flush=True)
                if t < best_t * 0.99:
                    best_c, best_t = c, t
            if best_c is not None:
                trial = dict(plan)
                trial[l] = best_c
                t2 = evaluate(trial, shape, reps=2)
                cur2 = evaluate(plan, shape, reps=2)
                if t2 < cur2 * 0.995:
                    plan = trial
                    changed = True
                    print(f"KEEP layer {l} -> cfg {best_c[0]} splits {best_c[1]}: {t2:.4f} ms (was {cur2:.4f}); plan {plan_str(plan)}", flush=True)
        if not changed:
            break
    print(f"FINAL plan [{plan_str(plan)}]: {evaluate(plan, shape, reps=3):.4f} ms against default {evaluate({}, shape, reps=3):.4f} ms", flush=True)


if __name__ == "__main__":
    main()
