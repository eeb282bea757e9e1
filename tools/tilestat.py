"""Development aid: reads the table a -DFRAY_TILESTAT build of k_whitted leaves (tools/tilestat_run.sh): per 8x8 tile
{start, end (s_memtime of the wave's XCD), wave, the wave's kernel start, rounds, cheap-step iterations, lane-rounds at a search, at a direct-light loop}.
XCDs do not share a time base, so times are taken relative to the wave's own kernel start."""
import sys
import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
idx = np.nonzero(t[:, 1] > 0)[0]
t = t[idx]
start = (t[:, 0] - t[:, 3]).astype(np.float64)
end = (t[:, 1] - t[:, 3]).astype(np.float64)
dur = end - start
span = end.max()
waves = len(np.unique(t[:, 2]))
print("tiles %d, waves %d, kernel span %.3g ticks; a tile: mean %.3g, median %.3g, p90 %.3g, p99 %.3g, max %.3g ticks (max = %.1f %% of the span)"
      % (len(t), waves, span, dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max(), 100 * dur.max() / span))
print("sum of tile time / (waves x span) = %.3f" % (dur.sum() / (waves * span)))
for k in range(10):
    x = (k + 0.5) / 10 * span
    print("  at %.2f of the span: %5d waves inside a tile" % ((k + 0.5) / 10, int(((start <= x) & (end > x)).sum())))
rounds = t[:, 4].astype(np.float64)
print("rounds per tile: mean %.1f, p90 %.0f, p99 %.0f, max %.0f; ticks per round: mean %.3g; lanes at a search per round %.1f, at a light loop %.1f"
      % (rounds.mean(), np.percentile(rounds, 90), np.percentile(rounds, 99), rounds.max(), dur.sum() / max(rounds.sum(), 1), t[:, 6].sum() / max(rounds.sum(), 1), t[:, 7].sum() / max(rounds.sum(), 1)))
order = np.argsort(-dur)[:12]
print("longest tiles: (tile, start/span, duration/span, rounds, cheap iterations, ticks/round, search lanes/round, light-loop lanes/round)")
for i in order:
    r = max(rounds[i], 1)
    print("   ", int(idx[i]), round(start[i] / span, 3), round(dur[i] / span, 3), int(rounds[i]), int(t[i, 5]), int(dur[i] / r), round(t[i, 6] / r, 1), round(t[i, 7] / r, 1))
