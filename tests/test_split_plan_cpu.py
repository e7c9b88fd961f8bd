"""Host logic of the opt-in split-operand modes (no GPU): which layers of the headline UNet plan run split GEMMs (ConvBnRelu._split3), which input
buffers get a device word for their largest magnitude from the passes that write them (Runner.plan_amax), and the switch itself."""
import torch

import pytorch_camvid_amd as A
from pytorch_camvid_amd import engine
from pytorch_camvid_amd.modules import runner_of


def _headline_plan(net):
    plan = engine.Plan(8, 3, 360, 480)
    plan.output = net._emit(plan, plan.input)
    plan.seal()
    return plan


def test_switch_values():
    net = A.UNet(3, 12)
    R = runner_of(net)
    assert engine.split_fmt(R) == 0                                # exact-fp32 MFMA is the default
    for v, want in ((2, 2), (3, 3), (0, 0), (True, 3)):
        R.w2d_split = v
        assert engine.split_fmt(R) == want
    A.set_split_operands(net, 2)
    assert R.w2d_split == 2
    try:
        A.set_split_operands(net, 5)
        assert False
    except ValueError:
        pass


def test_environment_switch_is_validated(monkeypatch, capsys):
    """CVK_W2D_SPLIT moves the fp32 products onto 16-bit MFMAs for every network of the process: a bad value is a ValueError naming the variable
    (it used to be a bare KeyError on the first forward pass), a non-zero value is announced once on stderr."""
    import pytest
    monkeypatch.delenv("CVK_W2D_SPLIT", raising=False)
    assert engine._split_mode_from_env() == 0
    for raw, want in (("0", 0), ("1", 3), ("3", 3), ("2", 2), (" 2 ", 2)):
        monkeypatch.setenv("CVK_W2D_SPLIT", raw)
        assert engine._split_mode_from_env() == want
    assert "CVK_W2D_SPLIT" in capsys.readouterr().err
    for raw in ("", "true", "4"):
        monkeypatch.setenv("CVK_W2D_SPLIT", raw)
        with pytest.raises(ValueError, match="CVK_W2D_SPLIT"):
            engine._split_mode_from_env()


def test_thirteen_layers_run_the_split_gemms_and_none_by_default():
    net = A.UNet(3, 12)
    plan = _headline_plan(net)
    R = runner_of(net)
    R.w2tile, R.w2tile_dgrad = R.tile_for(plan)
    convs = [op for op in plan.ops if isinstance(op, engine.ConvBnRelu)]
    assert len(convs) == 23
    assert not any(op._split3(R) for op in convs)
    R.w2d_split = 2
    split = [op for op in convs if op._split3(R)]
    assert len(split) == 13                                           # the channel-heavy layers of DESIGN.md 5b
    for op in split:
        assert (op.cout % 256 == 0 and op.src.ld % 128 == 0) or (op.src.ld % 256 == 0 and op.cout % 128 == 0)
        assert op.src.H <= 180                                        # none of the full-resolution 64-channel layers


def test_amax_blocks_are_planned_for_every_fusable_input():
    """Every conv block whose input is written only by BatchNorm-apply passes (incl. the fused pool), the bilinear upsampling (bounded by ITS
    input) or the zero frame gets a block; the network input (written by the caller) does not; nothing is planned outside the fp16 mode, in
    evaluation or without gradients."""
    net = A.UNet(3, 12)
    plan = _headline_plan(net)
    R = runner_of(net)
    R.w2tile, R.w2tile_dgrad = R.tile_for(plan)

    def planned(fmt, training=True, need_grad=True):
        R.w2d_split = fmt
        st = engine.RunState([], training, need_grad)
        st.device, st.plan = torch.device("cpu"), plan
        R.plan_amax(plan, st)
        return st
    for fmt, tr, ng in ((0, True, True), (3, True, True), (2, False, True), (2, True, False)):
        st = planned(fmt, tr, ng)
        assert st.amax == {} and st.amax_spare == []
    st = planned(2)
    convs = [op for op in plan.ops if isinstance(op, engine.ConvBnRelu)]
    assert plan.input.id not in st.amax
    nw = R.lib.cvk_amax_block_words()
    for op in convs:
        if op.src is plan.input:
            continue
        assert op.src.id in st.amax, op.src.name                      # all 22 other conv inputs of the UNet are fusable
        b = st.amax[op.src.id]
        assert b.dtype == torch.int32 and b.numel() == nw and int(b.abs().max()) == 0
    # bilinear outputs: their own block (a snapshot of the input's maximum is copied in: the consumer's scale must not move afterwards)
    ups = [op for op in plan.ops if isinstance(op, engine.Upsample)]
    assert len(ups) == 4
    for op in ups:
        assert op.dst.id in st.amax and op.src.id in st.amax and st.amax[op.dst.id].data_ptr() != st.amax[op.src.id].data_ptr()
    # distinct blocks, and one spare per conv block for the |dy| maxima of the backward pass
    ptrs = {t.data_ptr() for t in st.amax.values()} | {t.data_ptr() for t in st.amax_spare}
    assert len(ptrs) == len(st.amax) + len(st.amax_spare) and len(st.amax_spare) >= 22
