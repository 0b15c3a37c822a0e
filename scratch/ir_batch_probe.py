"""Where the IR batch tracker's time goes: device walk with / without statistics, association, host objects."""
import sys, time, json
sys.path.insert(0, "classifier-pipeline_amd")
import numpy as np, torch
from cpx.config import Config
from cpx.track.clip import Clip
from cpx.track.irtrackextractor import IRTrackExtractor
import cProfile, pstats, io

S, T, H, W = 64, 32, 480, 640
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(99)
scene = torch.randint(40, 200, (S, 1, H, W), generator=g, device=dev, dtype=torch.int16)
video = scene + torch.randint(-2, 3, (S, T, H, W), generator=g, device=dev, dtype=torch.int16)
for t in range(T):
    x0 = (11 * t) % (W - 60)
    video[:, t, 150:200, x0:x0 + 60] = 235
video = video.clamp_(0, 255).to(torch.uint8).permute(1, 0, 2, 3).contiguous()
tr = IRTrackExtractor(Config.get_defaults().tracking)
def run(stats):
    clips = []
    for v in range(S):
        c = Clip(tr.config, "ir-%d.mp4" % v, type="IR"); c.frames_per_second = 10; clips.append(c)
    t0 = time.perf_counter(); tr.parse_frames_batch(clips, video, calc_stats=stats); return time.perf_counter() - t0
run(True)
out = {"with_stats_s": min(run(True) for _ in range(3)), "without_stats_s": min(run(False) for _ in range(3))}
pr = cProfile.Profile(); pr.enable(); run(False); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25); out["profile"] = s.getvalue()
print(json.dumps({k: v for k, v in out.items() if k != "profile"})); print(out["profile"])
