"""The MOTS loop of bench.py (`configs.mots_loop`) alone, for `rocprofv3 --kernel-trace --stats`: N frames through OmniMOTSFrame.run_stream on
unicorn_track_large_mot_challenge_mask at 800x1280 (64 candidates per frame)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unicorn_amd.tracker import OmniMOTSFrame, QuasiDenseEmbedTracker
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
H, W = 800, 1280
dev = torch.device("cuda:0")
with torch.no_grad():
    mm = bench.Stream("unicorn_track_large_mot_challenge_mask", "f16x2", "mot", H, W, 1, dev, seed=1, corr_prec=2)
    o, _ = mm.model(mm.frames[1])
    sc = (o[0][0, :, 4] * o[0][0, :, 5]).sort(descending=True)[0]
    thr = float((sc[63] + sc[64]) / 2)
    kw = dict(init_score_thr=float(sc[16]), obj_score_thr=float(sc[40]), match_score_thr=0.5, memo_tracklet_frames=10, memo_backdrop_frames=1,
              memo_momentum=0.8, nms_conf_thr=float(sc[40]), nms_backdrop_iou_thr=0.3, nms_class_iou_thr=0.7, with_cats=True, match_metric="bisoftmax")
    mots = OmniMOTSFrame(mm.model, QuasiDenseEmbedTracker(**kw), (H, W), num_classes=1, confthre=thr, nmsthre=0.7, embed_score_thr=thr,
                         mask_thres=0.3, d_rate=mm.cfg.d_rate)
    for _ in mots.run_stream((mm.frames[1 + j % 4] for j in range(4)), (1080, 1920)):
        pass
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    n = sum(1 for _ in mots.run_stream((mm.frames[1 + j % 4] for j in range(N)), (1080, 1920)))
    torch.cuda.synchronize()
    print("frames", n, "ms per frame %.3f" % (1e3 * (time.perf_counter() - t0) / n))
