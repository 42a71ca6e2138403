"""Configuration read by the hot path.

In drop-in mode (under the reference's train.py) `sync_from_reference()` copies
the values the reference's global `cfg` (config.py:60-190) holds for these
fields after `assert_and_infer_cfg`; standalone (bench.py, tests) the defaults
below are the reference's defaults for the HRNet-OCR-MScale recipes.
"""


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _defaults():
    c = AttrDict()
    c.MODEL = AttrDict(
        BNFUNC=None,                 # config.py:216-225 picks the norm layer; None -> semseg_amd.nn.BatchNorm2d
        ALIGN_CORNERS=False,         # config.py:125
        MSCALE_LO_SCALE=0.5,         # config.py:126
        N_SCALES=None,               # config.py:124
        SEGATTN_BOT_CH=256,          # config.py:130
        ASPP_BOT_CH=256,             # config.py:131
        MSCALE_INNER_3x3=True,       # config.py:131
        HRNET_CHECKPOINT="",         # config.py:147 (empty: random init, no file needed)
        OCR=AttrDict(MID_CHANNELS=512, KEY_CHANNELS=256),   # config.py:157-158
        # cfg.MODEL.OCR_EXTRA, config.py:161-190 (HRNetV2-W48)
        OCR_EXTRA=AttrDict(
            STAGE1=AttrDict(NUM_MODULES=1, NUM_BLOCKS=[4], NUM_CHANNELS=[64], BLOCK="BOTTLENECK"),
            STAGE2=AttrDict(NUM_MODULES=1, NUM_BRANCHES=2, NUM_BLOCKS=[4, 4], NUM_CHANNELS=[48, 96], BLOCK="BASIC"),
            STAGE3=AttrDict(NUM_MODULES=4, NUM_BRANCHES=3, NUM_BLOCKS=[4, 4, 4], NUM_CHANNELS=[48, 96, 192], BLOCK="BASIC"),
            STAGE4=AttrDict(NUM_MODULES=3, NUM_BRANCHES=4, NUM_BLOCKS=[4, 4, 4, 4], NUM_CHANNELS=[48, 96, 192, 384], BLOCK="BASIC"),
        ),
    )
    c.LOSS = AttrDict(OCR_ALPHA=0.4, OCR_AUX_RMI=False, SUPERVISED_MSCALE_WT=0)   # config.py:150-155
    c.DATASET = AttrDict(NUM_CLASSES=19, IGNORE_LABEL=255)
    c.OPTIONS = AttrDict(INIT_DECODER=False)
    c.REDUCE_BORDER_EPOCH = -1       # config.py:60 (the 'scl-poly' LR schedule switches there)
    return c


cfg = _defaults()


def sync_from_reference(ref_cfg):
    """Copy the fields the hot path reads from the reference's global cfg."""
    m = ref_cfg.MODEL
    for k in ("ALIGN_CORNERS", "MSCALE_LO_SCALE", "N_SCALES", "SEGATTN_BOT_CH", "ASPP_BOT_CH", "MSCALE_INNER_3x3",
              "HRNET_CHECKPOINT"):
        if hasattr(m, k):
            cfg.MODEL[k] = getattr(m, k)
    cfg.MODEL.OCR.MID_CHANNELS = m.OCR.MID_CHANNELS
    cfg.MODEL.OCR.KEY_CHANNELS = m.OCR.KEY_CHANNELS
    for k in ("OCR_ALPHA", "OCR_AUX_RMI", "SUPERVISED_MSCALE_WT"):
        cfg.LOSS[k] = getattr(ref_cfg.LOSS, k)
    cfg.DATASET.NUM_CLASSES = ref_cfg.DATASET.NUM_CLASSES
    cfg.DATASET.IGNORE_LABEL = ref_cfg.DATASET.IGNORE_LABEL
    cfg.OPTIONS.INIT_DECODER = ref_cfg.OPTIONS.INIT_DECODER
    cfg.REDUCE_BORDER_EPOCH = getattr(ref_cfg, "REDUCE_BORDER_EPOCH", -1)
    assert not cfg.MODEL.ALIGN_CORNERS, "only align_corners=False is implemented"
    for k in ("MSCALE_OLDARCH", "MSCALE_DROPOUT", "MSCALE_CAT_SCALE_FLT"):      # config.py:132-135
        assert not getattr(m, k, False), "cfg.MODEL.%s is not on the accelerated path" % k
