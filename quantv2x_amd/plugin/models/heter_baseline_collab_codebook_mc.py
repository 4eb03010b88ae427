"""Multi-class codebook model; mirror of ``opencood/models/heter_baseline_collab_codebook_mc.py:30-57``."""
from .heter_baseline_collab_codebook import _CodebookMixin
from .heter_model_baseline_mc import HeterModelBaselineMC


class HeterBaselineCollabCodebookMC(_CodebookMixin, HeterModelBaselineMC):
    def __init__(self, args):
        super().__init__(args)
        self._build_codebook(args)
