"""`nets` package of the MI355X-native NBM detector: same constructors, module names, `state_dict`
keys and return structures as the reference's `nbm_model/nets` (reference nets/__init__.py:1-6)."""
from .nbm_model import build


def build_model(args):
    return build(args)
