"""Test helper: which scenes of a batch hold a ReLU unit whose pre-activation is within floating-point rounding
of zero in the CPU oracle's forward.

Why the backward tests need it.  Two correct fp32 implementations of the forward differ in the last bits of
every pre-activation (summation order: ~1e-7 relative).  A ReLU unit whose pre-activation lies inside that window
may be active in one implementation and inactive in the other; its forward contribution is ~1e-7 either way, but
its BACKWARD contribution is switched on or off as a whole, which changes gradients by a finite amount (percent
of max|grad| for an unlucky unit).  Scenes are independent, so the damage is confined to the gradients of the
scene that holds the unit.  A scene none of whose ReLU inputs is closer to zero than WINDOW is "clean": on clean
scenes two correct implementations must agree to fp32 rounding, and the tests gate there at 2e-5 of max|grad| (measured: <= 3.3e-6);
a wrong term (a missing self-loop factor, an indexing slip in a padded row) shows on clean scenes just the same.
"""
import contextlib

import torch

WINDOW = 2e-6      # several times the observed pre-activation differences (<= 5e-7 at these magnitudes)


class ReluProbe:
    def __init__(self, B: int):
        self.near = torch.full((B,), float("inf"))
        self.units = 0

    def see(self, x: torch.Tensor) -> None:
        B = self.near.shape[0]
        if x.dim() >= 2 and x.shape[0] % B == 0:      # (B, ...) or scene-major rows (B*N, ...)
            self.near = torch.minimum(self.near, x.detach().abs().reshape(B, -1).min(dim=1).values)
            self.units += x.numel() // B

    def clean(self) -> torch.Tensor:
        """(B,) bool: no ReLU input of the scene lies within WINDOW of zero."""
        return self.near > WINDOW


@contextlib.contextmanager
def relu_probe(B: int):
    """Records, per scene, the smallest |pre-activation| over every torch.relu call whose input is batch-leading
    with B scenes (all of the oracle's are)."""
    probe = ReluProbe(B)
    orig = torch.relu

    def spy(x):
        probe.see(x)
        return orig(x)
    torch.relu = spy
    try:
        yield probe
    finally:
        torch.relu = orig
