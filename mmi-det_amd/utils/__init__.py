"""Overlay package: the native hot-path modules (loss, general, torch_utils, autoanchor) first; every other `utils.*`
module of an MMI-Det checkout (datasets, plots, metrics, google_utils, wandb_logging ...) resolves to the reference's own
file (mmidet_hip/overlay.py)."""
from mmidet_hip import overlay as _overlay

_overlay.extend_package_path(__name__, __path__)
