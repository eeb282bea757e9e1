#!/usr/bin/env python3
"""What the ranks' own work allows an N-rank run to reach, measured on ONE GPU (run on the GPU box): for N in 2, 4, 8 every rank's
share of the frame (buckets r mod N, the partition bench.py --gpus N uses) is rendered alone and timed; the slowest share bounds the
N-rank frame.  predicted_efficiency = T(1) / (N x max share) -- a PREDICTION of the compute side only (no exchange, no second GPU
involved), not a scaling measurement.

  python tools/shard_balance.py OUT.json [workload ...]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import fray_amd
    from fray_amd import abi
    import bench
    out_path = sys.argv[1]
    names = sys.argv[2:] or ["cornell_pt64", "forest_dof256", "smallpt_pt64"]
    fray_amd.lib.frayhip_init(0)
    res = {"note": __doc__.split("\n\n")[0].replace("\n", " "), "workloads": {}}
    for name in names:
        scene_file, W, H, over, text = bench.WORKLOADS[name]
        s = bench.open_scene(fray_amd, scene_file, W, H, over)
        s.beginRender()
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")

        def timed(first, stride, reps=3):
            best = 1e30
            for _ in range(reps + 1):                     # the first call warms the workspace up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.render_device(frame.data_ptr(), seed=42, bucket_first=first, bucket_stride=stride, stream=torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 1e3)
            return best

        whole = timed(0, 1)
        rec = {"workload": text, "whole_frame_ms": whole, "ranks": {}}
        for n in (2, 4, 8):
            shares = [timed(r, n) for r in range(n)]
            rec["ranks"][str(n)] = {"share_ms": shares, "max_ms": max(shares), "mean_ms": sum(shares) / n, "max_over_mean": max(shares) / (sum(shares) / n),
                                    "predicted_efficiency": whole / (n * max(shares))}
        res["workloads"][name] = rec
        s.close()
        print(name, "whole %.2f ms" % whole, {n: "max/mean %.3f, predicted efficiency %.3f" % (r["max_over_mean"], r["predicted_efficiency"]) for n, r in rec["ranks"].items()}, flush=True)
    from tools.source_hash import source_hash
    res["source_hash"] = source_hash()
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
