"""The launch wrappers as a package of kernel-family modules (vcvits_amd/ops/): one namespace, one-way imports, and
`ops.replace` reaching the modules that call a function through their own globals."""
import ast
import os

from vcvits_amd import ops

HERE = os.path.dirname(os.path.abspath(ops.__file__))
ORDER = [m.__name__.rsplit(".", 1)[1] for m in ops.FAMILIES]


def test_package_namespace_is_the_union_of_the_family_modules():
    for m in ops.FAMILIES:
        for k, v in vars(m).items():
            if not k.startswith("__"):
                assert getattr(ops, k) is v, (m.__name__, k)
    # the switches are shared objects, not copies
    assert ops._USE_X3 is ops.core._USE_X3 is ops.conv._USE_X3
    assert ops.LAUNCH_COUNTS is ops.core.LAUNCH_COUNTS is ops.attention.LAUNCH_COUNTS
    assert ops.CAPTURING is ops.weights.CAPTURING


def test_family_modules_import_one_way():
    """a module imports only from modules before it in ops.FAMILIES (no cycles, no import-time order traps)"""
    for i, name in enumerate(ORDER):
        tree = ast.parse(open(os.path.join(HERE, name + ".py")).read())
        for n in tree.body:
            if isinstance(n, ast.ImportFrom) and n.level == 1 and n.module:
                assert n.module in ORDER[:i], "%s imports from %s" % (name, n.module)


def test_replace_reaches_every_module_that_calls_the_function():
    real = ops.next_seed
    fake = lambda: 7  # noqa: E731
    try:
        assert ops.replace("next_seed", fake) is real
        assert ops.next_seed is fake and ops.blocks.next_seed is fake and ops.attention.next_seed is fake
    finally:
        ops.replace("next_seed", real)
    assert ops.blocks.next_seed is real and ops.attention.next_seed is real and ops.next_seed is real
