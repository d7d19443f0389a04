"""CPU checks of the drop-in surface: the C ABI exports, the registry/config contract, error behaviour."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from uvltrack_amd import _native, build
    build.build()
    return _native.load()


def test_every_declared_symbol_is_exported(lib):
    from uvltrack_amd import _native
    hdr = open(os.path.join(ROOT, "include", "uvltrack_hip.h")).read()
    declared = set(re.findall(r"\b(uvl_[a-z0-9_]+)\s*\(", hdr)) - {"uvl_config", "uvl_inputs", "uvl_outputs", "uvl_model_t", "uvl_tuning"}
    assert declared, "header parse failed"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_native.EXPORTS), (declared ^ set(_native.EXPORTS))


def test_config_struct_layout_matches_header(lib):
    from uvltrack_amd import _native
    assert ctypes.sizeof(_native.UvlConfig) == 4 * (5 + 64 + 12)
    assert lib.uvl_version() >= 1


def test_tuning_struct_and_no_global_tuning_state(lib):
    """uvl_tuning is a plain struct of int32 (-1 = heuristic) owned by a handle or passed per call; the library holds no
    process-global tuning variable (no global mutable state besides the error string)."""
    import subprocess
    from uvltrack_amd import _native
    assert ctypes.sizeof(_native.UvlTuning) == 4 * len(_native.TUNING_FIELDS)
    t = _native.UvlTuning(gemm_cfg=11)
    assert t.gemm_cfg == 11 and t.attn_cfg == -1
    nf = len(_native.TUNING_FIELDS)
    raw = (ctypes.c_int32 * (nf + 1))(*range(nf + 1))
    lib.uvl_tuning_init(ctypes.cast(raw, ctypes.POINTER(_native.UvlTuning)))
    assert list(raw) == [-1] * nf + [nf]                       # exactly the struct, not a byte more
    hdr = open(os.path.join(ROOT, "include", "uvltrack_hip.h")).read()
    fields = re.search(r"typedef struct uvl_tuning \{\s*int32_t ([^;]+);", hdr).group(1).replace(" ", "").split(",")
    assert tuple(fields) == _native.TUNING_FIELDS
    syms = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    assert "g_tune" not in syms
    assert lib.uvl_tune_set(None, b"gemm_cfg", 3) < 0          # a handle is required


def test_library_is_stamped_with_the_tested_toolchain(lib, monkeypatch):
    """The build refuses another hipcc than build.TESTED_HIPCC (hand-counted waits), stamps the version into the library, and the
    binding refuses a library with another stamp -- both behind one environment override."""
    from uvltrack_amd import _native, build
    assert lib.uvl_build_toolchain().decode() == build.TESTED_HIPCC == build.hipcc_version("/opt/rocm/bin/hipcc")
    # the same release under another build hash is the same compiler; another version number, or an unstamped library, is not
    assert build.same_toolchain(build.TESTED_HIPCC.rsplit("-", 1)[0] + "-0123456789")
    assert not build.same_toolchain("HIP version: 7.3.0-" + build.TESTED_HIPCC.rsplit("-", 1)[1])
    assert not build.same_toolchain("unknown") and not build.same_toolchain("")
    monkeypatch.delenv(build.OVERRIDE_ENV, raising=False)
    monkeypatch.setattr(build, "TESTED_HIPCC", "HIP version: 0.0.0")
    with pytest.raises(RuntimeError, match="not the tested toolchain"):
        build.check_toolchain("/opt/rocm/bin/hipcc")
    monkeypatch.setattr(_native, "_lib", None)
    with pytest.raises(_native.NativeLibraryError, match="not by the tested toolchain"):
        _native.load()
    monkeypatch.setenv(build.OVERRIDE_ENV, "1")
    with pytest.warns(UserWarning):
        assert _native.load() is not None
    monkeypatch.setattr(_native, "_lib", None)              # the next load() sees the real constant again


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is about the no-GPU container")
    from uvltrack_amd import _native
    from uvltrack_amd.spec import spec_tiny
    cfg = _native.config_from_spec(spec_tiny())
    assert not lib.uvl_create(ctypes.byref(cfg))
    assert lib.uvl_last_error()


def test_bad_geometry_is_rejected(lib):
    from uvltrack_amd import _native
    from uvltrack_amd.spec import spec_tiny
    for kw in (dict(dim=96, heads=2), dict(head_dim=32), dict(text_len=80, max_pos=128)):
        cfg = _native.config_from_spec(spec_tiny(**kw))
        assert not lib.uvl_create(ctypes.byref(cfg))
        assert b"must" in lib.uvl_last_error() or b"bad" in lib.uvl_last_error()


def test_registry_and_build_model_surface():
    from lib import registry
    import lib.models as M
    from lib.config.uvltrack import config as C
    C.update_config_from_file(os.path.join(ROOT, "experiments", "uvltrack", "baseline_base.yaml"))
    assert set(["uvltrack"]) <= set(registry.MODELS)
    assert "modality_unified_feature_extractor" in registry.BACKBONES and "modality_adaptive_box_head" in registry.HEADS
    model = M.uvltrack.build_model(C.cfg)               # the access path tracking/profile_model.py:66-67 uses
    assert hasattr(model, "backbone") and hasattr(model, "box_head") and hasattr(model, "forward_test")
    from uvltrack_amd.spec import state_dict_schema
    sch = state_dict_schema(model.spec)
    sd = model.state_dict()
    assert set(sd) == set(sch) and all(tuple(sd[k].shape) == tuple(sch[k]) for k in sch)
    r = registry.Registry()
    r.register("a", 1)
    with pytest.raises(AssertionError):
        r.register("a", 2)


def test_unknown_yaml_key_raises_value_error(tmp_path):
    from lib.config.uvltrack import config as C
    p = tmp_path / "bad.yaml"
    p.write_text("MODEL:\n  NOT_A_KEY: 1\n")
    with pytest.raises(ValueError, match="not exist in config.py"):
        C.update_config_from_file(str(p))


def test_forward_test_on_cpu_tensors_raises():
    import torch
    import lib.models as M
    from lib.config.uvltrack import config as C
    from lib.utils.misc import NestedTensor
    from uvltrack_amd._native import NativeLibraryError
    C.update_config_from_file(os.path.join(ROOT, "experiments", "uvltrack", "baseline_base.yaml"))
    model = M.uvltrack.build_model(C.cfg)
    text = NestedTensor(torch.ones(1, 40).long(), torch.ones(1, 40).bool())
    with pytest.raises(NativeLibraryError):
        model.forward_test(torch.zeros(1, 3, 128, 128), torch.zeros(1, 3, 256, 256), text, torch.zeros(1, 3, 768), torch.ones(1).long())
