"""A stand-in ``taichi`` module that EXECUTES the reference's ``@ti.kernel`` bodies as plain Python.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- used in the build container, where
/root/reference is mounted, to let the reference's OWN K-Means code
(/root/reference/gsconverter/processing/gpu_ops.py:57-96 ``k_means_assign`` / ``k_means_update``,
driver ``_kmeans_taichi`` :178-191) produce golden labels and centroids
(oracle/make_golden_kmeans.py -> tests/golden/kmeans_ref.npz).  Taichi itself is not installable
here (no network).

What the shim implements -- exactly the subset of Taichi semantics those kernels use:

* ``ti.init`` / ``ti.sync``: no-ops; ``ti.gpu`` / ``ti.cpu``: tokens; ``ti.types.ndarray()``: an
  annotation object.  Kernel arguments are the caller's numpy arrays, used in place (Taichi copies
  them to the device and back, which is the same thing for a sequential execution).
* ``@ti.kernel``: the function's source is re-compiled after an AST rewrite that gives the Python
  body Taichi's default types (``default_fp = f32``, ``default_ip = i32``):
    - every float literal ``c`` becomes ``np.float32(c)`` and ``float(x)`` becomes ``np.float32(x)``,
      so all arithmetic on array elements (np.float32 scalars) stays in IEEE binary32 with one
      rounding per operation -- multiply then add, NOT fused.  (A real Taichi backend may contract
      ``dist += diff * diff`` into an FMA under its default fast-math; that is backend-dependent and
      invisible in the source, so the golden vectors carry the unfused source semantics and the GPU
      tests compare within the tolerance of SURVEY.md 8(c).)
    - ``ti.atomic_add(a[idx], v)`` becomes ``a[idx] += v``.  Taichi parallelises the outermost
      ``for i in range(N)``; the shim runs it sequentially, i.e. ONE of the orders the reference's
      f32 atomics can take (the reference is order-nondeterministic here).
  Kernels are rewritten lazily on their first call, so kernels that use features outside this
  subset (``sor_compute_mean_dists``: ``ti.Vector``, ``ti.floor``) can be defined without being run.
"""
from __future__ import annotations

import ast
import inspect
import sys
import textwrap
import types as _pytypes

import numpy as np


class _Rewrite(ast.NodeTransformer):
    """float literal -> np.float32(literal); float(x) -> np.float32(x); ti.atomic_add(t, v) -> t += v."""

    def visit_Constant(self, node):
        if isinstance(node.value, float):
            return ast.copy_location(
                ast.Call(func=ast.Attribute(value=ast.Name(id="__shim_np", ctx=ast.Load()), attr="float32", ctx=ast.Load()),
                         args=[node], keywords=[]), node)
        return node

    def visit_Call(self, node):
        self.generic_visit(node)
        if isinstance(node.func, ast.Name) and node.func.id == "float" and len(node.args) == 1:
            return ast.copy_location(
                ast.Call(func=ast.Attribute(value=ast.Name(id="__shim_np", ctx=ast.Load()), attr="float32", ctx=ast.Load()),
                         args=node.args, keywords=[]), node)
        return node

    def visit_Expr(self, node):
        c = node.value
        if (isinstance(c, ast.Call) and isinstance(c.func, ast.Attribute) and c.func.attr == "atomic_add"
                and isinstance(c.func.value, ast.Name) and c.func.value.id == "ti" and len(c.args) == 2):
            target = c.args[0]
            if not isinstance(target, ast.Subscript):
                raise NotImplementedError("taichi_shim: atomic_add on a non-subscript target")
            target.ctx = ast.Store()
            value = self.visit(c.args[1])
            return ast.copy_location(ast.AugAssign(target=target, op=ast.Add(), value=value), node)
        self.generic_visit(node)
        return node


def _compile_kernel(fn):
    src = textwrap.dedent(inspect.getsource(fn))
    tree = ast.parse(src)
    fdef = tree.body[0]
    assert isinstance(fdef, ast.FunctionDef)
    fdef.decorator_list = []
    for a in fdef.args.args:  # annotations like ti.types.ndarray() are irrelevant to a Python execution
        a.annotation = None
    tree = ast.fix_missing_locations(_Rewrite().visit(tree))
    glb = dict(fn.__globals__)
    glb["__shim_np"] = np
    code = compile(tree, filename="<taichi_shim:%s>" % fn.__name__, mode="exec")
    exec(code, glb)
    return glb[fdef.name]


def kernel(fn):
    state = {}

    def run(*args):
        if "f" not in state:
            state["f"] = _compile_kernel(fn)
        with np.errstate(all="ignore"):
            return state["f"](*args)

    run.__name__ = fn.__name__
    run.__wrapped__ = fn
    return run


def _make_module() -> _pytypes.ModuleType:
    m = _pytypes.ModuleType("taichi")
    m.__doc__ = __doc__
    m.__shim__ = True
    m.gpu, m.cpu = "gpu", "cpu"
    m.f32, m.i32 = np.float32, np.int32
    m.init = lambda *a, **k: None
    m.sync = lambda: None
    m.kernel = kernel
    types = _pytypes.ModuleType("taichi.types")
    types.ndarray = lambda *a, **k: "ndarray"
    m.types = types
    return m


def install() -> _pytypes.ModuleType:
    """Put the shim into sys.modules as ``taichi`` (idempotent); returns it."""
    cur = sys.modules.get("taichi")
    if cur is not None and getattr(cur, "__shim__", False):
        return cur
    if cur is not None:
        raise RuntimeError("a real taichi is already imported")
    m = _make_module()
    sys.modules["taichi"] = m
    sys.modules["taichi.types"] = m.types
    return m


def uninstall():
    for k in ("taichi", "taichi.types"):
        if getattr(sys.modules.get(k), "__shim__", False) or k == "taichi.types":
            sys.modules.pop(k, None)
