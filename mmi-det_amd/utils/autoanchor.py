"""Anchor-order check run once in Model.__init__ (reference: utils/autoanchor.py:12-20)."""


def check_anchor_order(m):
    """Flip the anchor tables of a Detect module if their area order disagrees with the stride order."""
    area = m.anchor_grid.prod(-1).view(-1)
    if (area[-1] - area[0]).sign() != (m.stride[-1] - m.stride[0]).sign():
        m.anchors[:] = m.anchors.flip(0)
        m.anchor_grid[:] = m.anchor_grid.flip(0)


# names this module does not define (the reference's helpers outside the hot path) come from the reference checkout's
# module of the same name when one is overlaid: mmidet_hip/overlay.py
from mmidet_hip.overlay import fall_through as _fall_through  # noqa: E402

__getattr__ = _fall_through(__name__)
