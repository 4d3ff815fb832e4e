"""Anchor-order check run once in Model.__init__ (reference: utils/autoanchor.py:12-20)."""


def check_anchor_order(m):
    """Flip the anchor tables of a Detect module if their area order disagrees with the stride order."""
    area = m.anchor_grid.prod(-1).view(-1)
    if (area[-1] - area[0]).sign() != (m.stride[-1] - m.stride[0]).sign():
        m.anchors[:] = m.anchors.flip(0)
        m.anchor_grid[:] = m.anchor_grid.flip(0)
