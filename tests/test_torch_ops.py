"""torch.ops.batrack_hip: the operator registration over the C ABI (batrack_amd/csrc/torch_ops.cpp) loads and carries the
declared schemas; argument checks raise before anything touches a device (no GPU needed)."""
import pytest
import torch

from batrack_amd import _lib


def test_operators_are_registered_with_their_schemas():
    ops = _lib.torch_ops(strict=True)
    for name in ("plan_create", "plan_destroy", "plan_info", "ba_step"):
        assert hasattr(ops, name)
    s = str(torch.ops.batrack_hip.ba_step.default._schema)
    assert "int plan" in s and "Tensor(a!) poses_out" in s and "Tensor(b!) patches_out" in s and "int phase" in s
    s = str(torch.ops.batrack_hip.plan_create.default._schema)
    assert s.startswith("batrack_hip::plan_create(Tensor ii, Tensor jj, Tensor kk, int n_buf, int p_tot, int fixedp")


def test_argument_checks_raise():
    ops = _lib.torch_ops(strict=True)
    i32 = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="int64"):
        ops.plan_create(i32, i32, i32, 2, 8, 1, 0, 0)
    f = torch.zeros(8)
    with pytest.raises(RuntimeError, match="workspace|GPU"):
        ops.ba_step(0, f, f, f, f, 1, f, f, 3, f, f, f, [0.0, 0.0, 1.0, 1.0], 1e-4, 10.0, 0.05, 1, False, 0)
