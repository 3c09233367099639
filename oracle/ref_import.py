"""TEST INFRASTRUCTURE ONLY — import the reference's `nbm_model.nets` in THIS container.

Used by `oracle/make_golden.py` and by the container-only tests that validate the
oracle restatement against the real reference (they skip when /root/reference is
absent, which is always the case on the GPU box).  Never imported by the product.

* torchvision is provided by `oracle/tv_standin.py`.
* `train.py` cannot be imported (tensorboard + script-relative imports), so the
  default config is obtained by extracting `get_args_parser` (reference
  nbm_model/train.py:21-168) with `ast` and evaluating it in isolation.
"""
import argparse
import ast
import os
import sys

REF_ROOT = os.environ.get('NBM_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'nbm_model', 'nets'))


def import_nets():
    """Returns the reference package `nbm_model.nets` (module object)."""
    from . import tv_standin
    tv_standin.install()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import importlib
    return importlib.import_module('nbm_model.nets')


def default_args(device='cpu', **overrides):
    """Namespace with the reference's train.py defaults + setattr_others() derived fields."""
    src = open(os.path.join(REF_ROOT, 'nbm_model', 'train.py')).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'get_args_parser'][0]
    ns = {'argparse': argparse}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'train.py:get_args_parser', 'exec'), ns)
    args = ns['get_args_parser']().parse_args([])
    args.device = device
    for k, v in overrides.items():
        setattr(args, k, v)
    nets = import_nets()
    from nbm_model.nets.util.nets_utils import setattr_others
    setattr_others(args)
    return args


def build_reference_model(args, train=False):
    nets = import_nets()
    model, criterion = nets.build_model(args)
    model.train(train)
    return model, criterion
