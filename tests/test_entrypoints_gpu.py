"""The two command-line entry points on a real device, run as the user / the driver runs them (subprocesses of this repository's
root): `tracking/profile_model.py` -- the drop-in of the reference tool (/root/reference/tracking/profile_model.py:30-47,49-85; its two
result lines are the contract, :46-47) -- and `bench.py` through its own launcher with an RCCL process group of one rank."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "%s\n--- stdout ---\n%s\n--- stderr ---\n%s" % (" ".join(cmd), r.stdout[-3000:], r.stderr[-3000:])
    return r.stdout


@pytest.mark.parametrize("config,extra", [("baseline_base", []), ("baseline_base", ["--mode", "NLBBOX", "--batch", "2"]), ("baseline_large", ["--mode", "BBOX"])])
def test_profile_model_cli(config, extra):
    """registry -> build_model(cfg from the reference's own yaml) -> load_state_dict(strict) -> .to(device).eval() -> forward_test loop."""
    out = _run([sys.executable, os.path.join("tracking", "profile_model.py"), "--script", "uvltrack", "--config", config, "--iters", "3", "10"] + extra)
    lat = re.search(r"^The average overall latency is ([0-9.]+) ms$", out, re.M)
    fps = re.search(r"^FPS is ([0-9.]+) fps$", out, re.M)
    assert lat and fps, out
    batch = int(extra[extra.index("--batch") + 1]) if "--batch" in extra else 1
    ms, f = float(lat.group(1)), float(fps.group(1))
    assert 0.05 < ms < 1000.0
    assert abs(f - batch * 1000.0 / ms) <= 0.02 * f + 0.5          # the two lines describe the same measurement


def test_profile_model_sustains_the_engine_step_rate():
    """The north-star entry point's own number: `tracking/profile_model.py --config baseline_base` with the reference's default 500 / 1000
    iterations drives nn.Module.forward_test (engine lookup, input canonicalisation, output dict per call); it must sustain >= 0.9 x the rate
    of the pre-validated engine step bench.py times (HipEngine.make_eager_step) on the same workload in this test process -- i.e. the drop-in
    surface is not host-bound.  (The measured lines are committed in profiles/r04_profile_model.txt.)"""
    import time
    import numpy as np
    import torch
    out = _run([sys.executable, os.path.join("tracking", "profile_model.py"), "--script", "uvltrack", "--config", "baseline_base"])
    fps_cli = float(re.search(r"^FPS is ([0-9.]+) fps$", out, re.M).group(1))
    sys.path.insert(0, ROOT)
    from lib.config.uvltrack import config as Cfg
    from uvltrack_amd import weightgen as wg
    from uvltrack_amd.engine import HipEngine
    from uvltrack_amd.spec import spec_from_cfg
    Cfg.update_config_from_file(os.path.join(ROOT, "experiments", "uvltrack", "baseline_base.yaml"))
    spec = spec_from_cfg(Cfg.cfg)
    dev = torch.device("cuda:0")
    eng = HipEngine(spec, dev, max_batch=1)
    eng.load_state_dict(wg.make_state_dict(spec, 0, include_unused=False))
    inp = wg.make_inputs(spec, batch=1, seed=0, flags=[1])          # flag 1 (NL): what the CLI runs without --mode
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    step = eng.make_eager_step(t(inp["template"]), t(inp["search"]), t(inp["ids"]), t(inp["mask"]), t(inp["prompt"]), t(inp["flag"]), outs=eng.alloc_outputs(1))
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        step()
    torch.cuda.synchronize()
    fps_step = 1000.0 / (time.perf_counter() - t0)
    assert fps_cli >= 0.9 * fps_step, "profile_model.py %.1f fps against %.1f fps of the engine step" % (fps_cli, fps_step)


def test_bench_self_launch_one_rank_rccl():
    """BASELINE configs[4]'s per-GPU shard through the launcher the multi-GPU runs use: `bench.py --gpus 1 --dist` re-launches itself
    under torch.distributed.run, joins an RCCL ("nccl") process group of ONE rank and all-gathers the boxes every step -- the same code
    as N ranks.  (No N > 1 run has been possible: one GPU per box.)"""
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--dist", "--model", "L", "--batch", "8", "--steps", "3", "--warmup", "2", "--blocks", "1",
                "--no-cpu-baseline"])
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1
    assert line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp1"
    assert line["value"] > 0 and line["outputs_finite"] and line["scaling"] == "weak"
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line


def test_bench_default_line_has_the_contract_fields():
    out = _run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--blocks", "3", "--no-batched", "--cpu-frames", "1"])
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["rccl_ranks"] == 0 and line["steps"] == 20 and line["vs_baseline"] is None
    r = line["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["regime"] in ("mfma", "hbm", "latency") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    assert line["cpu_baseline"]["kind"] == "port-torch" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["numpy_port"]["value"] > 0
