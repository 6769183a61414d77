"""``Projector`` / ``LumpedProject`` names of reference src/oasisx/function.py.

Only reached by the reference when ``rotational=True`` (fracstep.py:237-251,593-602), which
no benchmark configuration uses; SURVEY.md lists it as a "next" row.  The names exist so that
``from oasisx_amd import Projector`` works; constructing one raises until the row is built.
"""
from __future__ import annotations

__all__ = ["Projector", "LumpedProject"]


class Projector:
    def __init__(self, function, space, bcs=None, petsc_options=None, jit_options=None,
                 form_compiler_options=None, metadata=None):
        raise NotImplementedError("Projector (L2 projection, rotational pressure update) is not "
                                  "implemented on the HIP path yet")


class LumpedProject:
    """Projector using a lumped mass matrix (raises in the reference too, function.py:146-153)."""

    def __init__(self):
        raise NotImplementedError
