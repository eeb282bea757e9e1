"""Development aid (GPU box): work counters and time of single 48x48 buckets of a Whitted frame -- what do the rays of an expensive bucket do?
   python tools/bucket_stats.py scenes/hw9/dragon.fray 1920 1080 844 500 100"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fray_amd

path, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
s = fray_amd.Scene.parseScene(path)
s.settings.frameWidth, s.settings.frameHeight, s.settings.wantAA = W, H, 0
s.beginRender(0)
for b in map(int, sys.argv[4:]):
    s.render(bucket_first=b, bucket_stride=1 << 20)
    _, st = s.render(bucket_first=b, bucket_stride=1 << 20, stats=True)
    _, st2 = s.render(bucket_first=b, bucket_stride=1 << 20)
    rays = st["closest_rays"] + st["shadow_rays"]
    print("bucket %d: %.3f ms; closest %d, shadow %d; per ray: node tests %.1f, KD inner visits %.1f, leaf triangle refs %.1f, triangle tests %.1f"
          % (b, st2["ms_kernels"], st["closest_rays"], st["shadow_rays"], st["node_tests"] / rays, st["kd_inner_visits"] / rays, st["leaf_refs"] / rays, st["tri_tests"] / rays))
