"""semseg_amd -- MI355X-native hot path of NVIDIA/semantic-segmentation
(HRNet-OCR-MScale forward/backward, losses, SyncBN, DDP) behind the reference's
own Python entry points.  Arithmetic lives in csrc/*.hip (libsemseg_hip.so)."""
__version__ = "0.1.0"
