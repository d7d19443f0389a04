"""Speed test of the per-frame forward pass -- drop-in for reference tracking/profile_model.py.

    python tracking/profile_model.py --script uvltrack --config baseline_base

Same flags, same input recipe (profile_model.py:70-74: randn template/search, ids = 1, mask = randn > 0.5,
randn prompt, flag = 1) and the same two output lines.  Extra flags: --mode, --batch, --seed, --iters.
Deviation (documented in DESIGN.md): the model runs with eval semantics; the reference script forgets
`.eval()`, leaving BERT dropout and BatchNorm in training mode.  Weights are synthetic (no checkpoints offline).
"""
import argparse
import importlib
import os
import sys
import time

prj_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
if prj_path not in sys.path:
    sys.path.append(prj_path)

import torch  # noqa: E402
from lib.utils.misc import NestedTensor  # noqa: E402


def parse_args():
    parser = argparse.ArgumentParser(description='Speed test of UVLTrack.forward_test')
    parser.add_argument('--script', type=str, default='uvltrack', choices=['uvltrack'], help='model script name')
    parser.add_argument('--config', type=str, default='baseline_base', help='yaml configure file name')
    parser.add_argument('--mode', type=str, default=None, choices=['BBOX', 'NL', 'NLBBOX'],
                        help='flag 0/1/2; default = the reference script default (flag 1, NL)')
    parser.add_argument('--batch', type=int, default=1)
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--iters', type=int, nargs=2, default=[500, 1000], metavar=('WARMUP', 'TIMED'))
    return parser.parse_args()


def evaluate_speed(model, template, search, text, prompt, flag, T_w=500, T_t=1000):
    '''Speed Test (reference loop: sync only before the warm-up and after the timed loop)'''
    print("testing speed ...")
    torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(T_w):
            _ = model.forward_test(template, search, text, prompt, flag)
        start = time.time()
        for _ in range(T_t):
            _ = model.forward_test(template, search, text, prompt, flag)
        torch.cuda.synchronize()
        end = time.time()
        avg_lat = (end - start) / T_t
        print("The average overall latency is %.2f ms" % (avg_lat * 1000))
        print("FPS is %.2f fps" % (1. / avg_lat * template.shape[0]))


if __name__ == "__main__":
    device = "cuda:0"
    torch.cuda.set_device(device)
    args = parse_args()
    yaml_fname = os.path.join(prj_path, 'experiments/%s/%s.yaml' % (args.script, args.config))
    config_module = importlib.import_module('lib.config.%s.config' % args.script)
    cfg = config_module.cfg
    config_module.update_config_from_file(yaml_fname)
    bs = args.batch
    z_sz = cfg.TEST.TEMPLATE_SIZE
    x_sz = cfg.TEST.SEARCH_SIZE
    dim = cfg.MODEL.HIDDEN_DIM

    model_module = importlib.import_module('lib.models')
    model = model_module.uvltrack.build_model(cfg)
    from uvltrack_amd import weightgen
    sd = weightgen.make_state_dict(model.spec, args.seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    torch.manual_seed(args.seed)
    template = torch.randn(bs, 3, z_sz, z_sz)
    search = torch.randn(bs, 3, x_sz, x_sz)
    text = NestedTensor(torch.ones(bs, 40).long(), torch.randn(bs, 40) > 0.5)
    prompt = torch.randn(bs, 3, dim)
    flag = torch.full((bs,), {None: 1, 'BBOX': 0, 'NL': 1, 'NLBBOX': 2}[args.mode]).long()
    model = model.to(device)
    model.eval()
    template, search, text, prompt, flag = template.to(device), search.to(device), text.to(device), prompt.to(device), flag.to(device)
    evaluate_speed(model, template, search, text, prompt, flag, *args.iters)
