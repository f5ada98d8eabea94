"""Mirror of /root/reference/label_anything/models/__init__.py for the hot path (on-path names only)."""
from labelanything_amd.models import (  # noqa: F401
    LabelAnything, Lam, build_lam, build_lam_no_vit, build_lam_vit_b, build_lam_vit_l, build_lam_vit_mae_b,
)
from labelanything_amd.config import ENCODER_SPECS as ENCODERS  # noqa: F401

model_registry = {
    "lam": build_lam,
    "lam_no_vit": build_lam_no_vit,
    "lam_b": build_lam_vit_b,
    "lam_l": build_lam_vit_l,
    "lam_mae_b": build_lam_vit_mae_b,
}
