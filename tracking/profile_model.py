"""Latency / FPS of the per-frame forward pass from the command line.

    python tracking/profile_model.py --script uvltrack --config baseline_base

CLI contract of the reference tool of the same name (tracking/profile_model.py:16-28 flags, :70-74 input recipe, :46-47 the
two result lines), so scripts that parse its output keep working.  What differs and why:
  * the model is put in eval mode (the reference never calls `.eval()`, which leaves BERT dropout and the BatchNorm of the
    head in training mode -- see DESIGN.md);
  * weights come from the deterministic generator (no checkpoints offline), inputs from a seeded torch generator;
  * `--mode`, `--batch`, `--seed`, `--iters WARMUP TIMED` are additions; without `--mode` the flag is 1 (NL) as in the reference.
"""
import argparse
import importlib
import os
import sys
import time

REPO = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
sys.path.insert(0, REPO) if REPO not in sys.path else None

import torch  # noqa: E402

FLAG_OF_MODE = {None: 1, "BBOX": 0, "NL": 1, "NLBBOX": 2}
TEXT_LEN = 40


def cli():
    ap = argparse.ArgumentParser(description="Speed test of UVLTrack.forward_test on the HIP path")
    ap.add_argument("--script", default="uvltrack", choices=["uvltrack"], help="model family (lib/config/<script>, experiments/<script>)")
    ap.add_argument("--config", default="baseline_base", help="yaml name under experiments/<script>/")
    ap.add_argument("--mode", default=None, choices=["BBOX", "NL", "NLBBOX"], help="modality flag 0 / 1 / 2 (default: 1 like the reference)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--iters", type=int, nargs=2, default=(500, 1000), metavar=("WARMUP", "TIMED"))
    return ap.parse_args()


def load_cfg(script: str, config: str):
    mod = importlib.import_module("lib.config.%s.config" % script)
    mod.update_config_from_file(os.path.join(REPO, "experiments", script, config + ".yaml"))
    return mod.cfg


def build(cfg, script: str, seed: int, device):
    from uvltrack_amd import weightgen
    model = getattr(importlib.import_module("lib.models"), script).build_model(cfg)
    weights = weightgen.make_state_dict(model.spec, seed)
    model.load_state_dict({name: torch.from_numpy(w) for name, w in weights.items()}, strict=True)
    return model.to(device).eval()


def synthetic_frame(cfg, batch: int, mode, seed: int, device):
    """randn crops, all-ones token ids, ~31 % valid text mask, randn prompt -- the reference recipe, seeded."""
    from lib.utils.misc import NestedTensor
    g = torch.Generator().manual_seed(seed)
    z, x, d = cfg.TEST.TEMPLATE_SIZE, cfg.TEST.SEARCH_SIZE, cfg.MODEL.HIDDEN_DIM
    frame = dict(
        template=torch.randn(batch, 3, z, z, generator=g),
        search=torch.randn(batch, 3, x, x, generator=g),
        text=NestedTensor(torch.ones(batch, TEXT_LEN, dtype=torch.long), torch.randn(batch, TEXT_LEN, generator=g) > 0.5),
        prompt=torch.randn(batch, 3, d, generator=g),
        flag=torch.full((batch,), FLAG_OF_MODE[mode], dtype=torch.long),
    )
    return {k: v.to(device) for k, v in frame.items()}


def time_forward(model, frame, warmup: int, timed: int) -> float:
    """Seconds per forward_test call; the device is synchronised only around the whole loop, like the reference."""
    call = lambda: model.forward_test(frame["template"], frame["search"], frame["text"], frame["prompt"], frame["flag"])
    with torch.no_grad():
        torch.cuda.synchronize()
        for _ in range(warmup):
            call()
        t0 = time.time()
        for _ in range(timed):
            call()
        torch.cuda.synchronize()
        return (time.time() - t0) / timed


def main():
    args = cli()
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    cfg = load_cfg(args.script, args.config)
    model = build(cfg, args.script, args.seed, device)
    frame = synthetic_frame(cfg, args.batch, args.mode, args.seed, device)
    print("testing speed ...")
    latency = time_forward(model, frame, *args.iters)
    print("The average overall latency is %.2f ms" % (latency * 1000))
    print("FPS is %.2f fps" % (args.batch / latency))


if __name__ == "__main__":
    main()
