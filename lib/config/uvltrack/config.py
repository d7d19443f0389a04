"""`cfg` defaults + strict yaml override with the reference's contract (lib/config/uvltrack/config.py):
  * `cfg` is a module-level attribute dict; `update_config_from_file(path)` overrides it in place
  * a yaml key that does not exist in the defaults raises ValueError("<key> not exist in config.py") (:169-187)
The default tree below carries every key of the reference's defaults so its own yaml files load unchanged;
only the MODEL/DATA/TEST keys named in uvltrack_amd.spec.spec_from_cfg influence the forward pass.
"""
import yaml


class AttrDict(dict):
    """dict with attribute access, recursive (stands in for easydict.EasyDict, which is not a dependency here)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


_DEFAULTS = yaml.safe_load("""
MODEL:
  HIDDEN_DIM: 384
  NUM_OBJECT_QUERIES: 1
  POSITION_EMBEDDING: sine
  PREDICT_MASK: false
  LEARNABLE_POSITION: false
  BACKBONE:
    TYPE: mae_vit
    DROP_PATH_RATE: 0.0
    PRETRAINED_PATH: ''
    FUSION_LAYER: [8, 9, 10, 11]
    CONT_LOSS_LAYER: [4, 5, 6, 7, 8, 9, 10, 11]
    TXT_TOKEN_MODE: token
    LANGUAGE:
      IMPLEMENT: pytorch
      TYPE: bert-base-uncased
      PATH: pretrained/bert/bert-base-uncased.tar.gz
      VOCAB_PATH: pretrained/bert/bert-base-uncased-vocab.txt
      BERT: {LR: 0.0001, ENC_NUM: 12, HIDDEN_DIM: 256, MAX_QUERY_LEN: 40}
  HEAD:
    TYPE: anchor_free
    HEAD_DIM: 384
    CLS_TOKENIZE: true
    OFFSET_SIGMOID: true
    JOINT_CLS: false
    DROP: 0.0
    SOFTMAX_ONE: false
    GROUNDING_DILATION: 1
    CONTRASTIVE_CONV: false
TRAIN:
  POSITIVE_MODE: ctr
  MODE: grounding
  VLTVG_AUG: false
  GROUNDING_RATIO: null
  VL_RATIO: null
  LR: 0.0001
  WEIGHT_DECAY: 0.0001
  EPOCH: 500
  LR_DROP_EPOCH: 400
  BATCH_SIZE: 16
  NUM_WORKER: 8
  OPTIMIZER: ADAMW
  BACKBONE_MULTIPLIER: 0.1
  GIOU_WEIGHT: 2.0
  L1_WEIGHT: 5.0
  AUX_WEIGHT: 0.0
  CONT_WEIGHT: 1.0
  CIB_WEIGHT: 0.01
  CTR_RATIO: 0.75
  DEEP_SUPERVISION: false
  FREEZE_STAGE0: false
  PRINT_INTERVAL: 50
  VAL_EPOCH_INTERVAL: 20
  GRAD_CLIP_NORM: 0.1
  DYNAMIC_CLS: false
  REDUCTION: sum
  GAUSSIAN_IOU: 0.3
  SCHEDULER: {TYPE: step, DECAY_RATE: 0.1, WARM_EPOCH: 30, MILESTONES: [200, 250, 290], GAMMA: 0.1}
DATA:
  CONTEXT_GAP: null
  MEAN: [0.485, 0.456, 0.406]
  STD: [0.229, 0.224, 0.225]
  MAX_SAMPLE_INTERVAL: 200
  TRAIN: {DATASETS_NAME: [GOT10K_vottrain], DATASETS_RATIO: [1], SAMPLE_PER_EPOCH: 60000}
  VAL: {DATASETS_NAME: [GOT10K_votval], DATASETS_RATIO: [1], SAMPLE_PER_EPOCH: 10000}
  VALTRACK: {DATASETS_NAME: [OTB99_test], DATASETS_RATIO: [1], SAMPLE_PER_EPOCH: 10000}
  VALVL: {DATASETS_NAME: [OTB99_test], DATASETS_RATIO: [1], SAMPLE_PER_EPOCH: 10000}
  SEARCH: {SIZE: 320, FACTOR: 5.0, NUMBER: 1, CENTER_JITTER: 4.5, SCALE_JITTER: 0.5, CENTER_JITTER_GROUNDING: 4.5, SCALE_JITTER_GROUNDING: 0.5}
  TEMPLATE: {SIZE: 128, FACTOR: 2.0, NUMBER: 1, CENTER_JITTER: 0, SCALE_JITTER: 0}
TEST:
  MODE: NL
  TEMPLATE_FACTOR: 2.0
  TEMPLATE_SIZE: 128
  SEARCH_FACTOR: 5.0
  SEARCH_SIZE: 320
  EPOCH: 500
  THRESHOLD: 0.5
  THRESHOLD_CONT: 0.0
  THRESHOLD_CLS: 0.0
  WINDOW_INFLUENCE: 0.49
  UPDATE_INTERVAL: 100000
  UPDATE_INTERVALS: {LASOT: [200], GOT10K_TEST: [200], TRACKINGNET: [200], VOT20: [200], VOT20LT: [200]}
""")

cfg = AttrDict(_DEFAULTS)


def _merge(base, override, path=""):
    for k, v in override.items():
        if k not in base:
            raise ValueError("{} not exist in config.py".format(k))
        if isinstance(v, dict) and isinstance(base[k], dict):
            _merge(base[k], v, path + k + ".")
        else:
            base[k] = v


def update_config_from_file(filename):
    with open(filename) as f:
        exp = yaml.safe_load(f) or {}
    _merge(cfg, exp)


def gen_config(config_file):
    def plain(d):
        return {k: plain(v) if isinstance(v, dict) else v for k, v in d.items()}
    with open(config_file, "w") as f:
        yaml.dump(plain(cfg), f, default_flow_style=False)
