"""Checkpoint -> flat weights file for hosts without Python / torch (include/unicorn_hip.h: uni_weights_file_cfg / uni_ctx_load_file; the role
tools/export_torchscript.py has in the reference).      python tools/export_weights.py --ckpt latest_ckpt.pth --exp unicorn_track_large --out large.uniw
With --synthetic the weights of oracle/synth.py are written instead of a checkpoint (what tests / tools/capi_host_demo use offline).  CPU only."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", required=True, help="experiment name, e.g. unicorn_track_large / unicorn_track_tiny_mask")
    ap.add_argument("--ckpt", default=None, help="released checkpoint ({'model': state_dict}, tools/track.py:186-188)")
    ap.add_argument("--synthetic", action="store_true", help="write the synthetic weights of oracle/synth.py (no checkpoint needed)")
    ap.add_argument("--precision", default="f16x2", choices=["f16x2", "fp32", "bf16"], help="default operand format recorded in the file (the host may override it)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args(argv)
    import torch
    from unicorn_amd.utils.checkpoint import export_flat
    if args.synthetic:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import synth
        import unicorn_oracle as uo
        sd = synth.synth_state_dict(uo.CONFIGS[args.exp])
    else:
        if not args.ckpt:
            ap.error("--ckpt or --synthetic")
        try:
            ck = torch.load(args.ckpt, map_location="cpu", weights_only=True)
        except Exception:      # noqa: BLE001
            ck = torch.load(args.ckpt, map_location="cpu", weights_only=False)
        sd = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
    n = export_flat(sd, args.exp, args.out, precision=args.precision)
    print("wrote %s: %d tensors, %.1f MB" % (args.out, n, os.path.getsize(args.out) / 1e6))
    return 0


if __name__ == "__main__":
    sys.exit(main())
