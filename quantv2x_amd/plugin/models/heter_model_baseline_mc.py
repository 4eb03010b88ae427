"""Multi-class (3-head) variant; mirror of ``opencood/models/heter_model_baseline_mc.py``
(head widths ``cls = A*C*C``, ``reg = 7*A*C``, ``dir = bins*A*C``, reference ``:137-141``)."""
from .heter_model_baseline import HeterModelBaseline


class HeterModelBaselineMC(HeterModelBaseline):
    @staticmethod
    def _head_widths(args):
        a, c = args['anchor_number'], args['num_class']
        return a * c * c, 7 * a * c, args['dir_args']['num_bins'] * a * c
