"""cilrs_mi355 -- MI355X-native engine behind the reference's CILRS nn.Module boundary
(model/autonomous_drive.py:361-399, notebook/notebook.ipynb:440-555)."""
from .model import CILRS, CILRSResNet50  # noqa: F401
from .train import CONFIG_A, CONFIG_B, LOSS_KEYS, TrainConfig, Trainer  # noqa: F401
