"""Cached inputs / targets / output gradients of one layer or block for the reconstruction loop (SURVEY.md §8(f)-4;
reference: quant/data_utill.py:13-264 — the file name keeps the reference's spelling so imports stay drop-in).

``save_inout`` runs the calibration set through the model in mini-batches with a forward hook on the target that stops
the forward as soon as the target has produced its output: targets come from the FP model; with ``asym`` the inputs are
re-captured with everything before the target quantised (BRECQ's asymmetric reconstruction).  ``save_grad`` captures
|∂KL/∂output| + 1 for the Fisher-weighted losses.
"""
import logging
from typing import Dict, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from .quant_block import BaseQuantBlock, QuantBasicTransformerBlock
from .quant_layer import QuantLayer
from .quant_model import QuantModel

logger = logging.getLogger(__name__)


class StopForwardException(Exception):
    """Raised by the hook to abandon the rest of the UNet once the target's tensors are captured."""


class DataSaverHook:
    """Forward hook (``with_kwargs=True``) that keeps the target's inputs and / or output."""

    def __init__(self, store_input: bool = False, store_output: bool = False, stop_forward: bool = False) -> None:
        self.store_input, self.store_output, self.stop_forward = store_input, store_output, stop_forward
        self.input_store = None
        self.output_store = None

    def __call__(self, module: nn.Module, input_batch: Tuple, kwargs: Dict, output_batch) -> None:
        if self.store_input:
            inputs = input_batch[:-1] if isinstance(input_batch[-1], int) else input_batch   # a trailing `split` int
            if isinstance(module, QuantBasicTransformerBlock):
                # the text context reaches a transformer block as a keyword (or second positional) argument
                ctx = kwargs.get("encoder_hidden_states", inputs[1] if len(inputs) > 1 else None)
                inputs = (inputs[0], ctx)
            self.input_store = inputs
        if self.store_output:
            self.output_store = output_batch
        if self.stop_forward:
            raise StopForwardException


def _run(model, xs, ts, cs, device):
    if cs is not None:
        return model(xs.to(device), ts.to(device), cs.to(device))
    return model(xs.to(device), ts.to(device))


class GetLayerInpOut:
    def __init__(self, model: QuantModel, layer: Union[QuantLayer, BaseQuantBlock], device: torch.device,
                 asym: bool = False, use_aq: bool = False) -> None:
        self.model, self.layer, self.device, self.asym, self.use_aq = model, layer, device, asym, use_aq
        self.data_saver = DataSaverHook(store_input=True, store_output=True, stop_forward=True)

    def __call__(self, xs: torch.Tensor, ts: torch.Tensor, cs: torch.Tensor = None):
        self.model.eval()
        self.model.set_quant_state(False, False)
        handle = self.layer.register_forward_hook(self.data_saver, with_kwargs=True)
        with torch.no_grad():
            try:
                _run(self.model, xs, ts, cs, self.device)
            except StopForwardException:
                pass
            if self.asym:
                # inputs as the quantised network produces them; the FP output captured above stays the target
                self.data_saver.store_output = False
                self.model.set_quant_state(use_wq=True, use_aq=self.use_aq)
                try:
                    _run(self.model, xs, ts, cs, self.device)
                except StopForwardException:
                    pass
                self.data_saver.store_output = True
        handle.remove()
        self.model.set_quant_state(False, False)
        self.layer.set_quant_state(True, self.use_aq)
        self.model.train()
        inputs = tuple(x.detach() for x in self.data_saver.input_store)
        out = self.data_saver.output_store
        outputs = (out.detach(),) if torch.is_tensor(out) else tuple(x.detach() for x in out)
        return inputs, outputs


def save_inout(model: QuantModel, layer: Union[QuantLayer, BaseQuantBlock], cali_data: Tuple[torch.Tensor], asym: bool = False,
               use_act: bool = False, batch_size: int = 128, keep_gpu: bool = True):
    """(cached_inputs tuple, cached_outputs) of `layer` over the whole calibration set (data_utill.py:13-52).  Chunks are
    parked on the host while the set is traversed (the reference does the same: the capture forwards need the device
    memory) and moved back to the device at the end when ``keep_gpu``."""
    device = next(model.parameters()).device
    get_inout = GetLayerInpOut(model, layer, device, asym, use_act)
    ins, outs = None, None
    for i in range(0, cali_data[0].size(0), batch_size):
        ipts, opts = get_inout(*(t[i: i + batch_size] for t in cali_data))
        if ins is None:
            ins, outs = tuple([] for _ in ipts), tuple([] for _ in opts)
        for store, t in zip(ins, ipts):
            store.append(t.cpu())
        for store, t in zip(outs, opts):
            store.append(t.cpu())
    cached_inputs = tuple(torch.cat(x) for x in ins)
    cached_outputs = tuple(torch.cat(x) for x in outs)
    if keep_gpu:
        cached_inputs = tuple(x.to(device) for x in cached_inputs)
        cached_outputs = tuple(x.to(device) for x in cached_outputs)
    for i, x in enumerate(cached_inputs):
        logger.info("input %d shape: %s", i, tuple(x.shape))
    for i, x in enumerate(cached_outputs):
        logger.info("output %d shape: %s", i, tuple(x.shape))
    return cached_inputs, (cached_outputs[0] if len(cached_outputs) == 1 else cached_outputs)


class GradSaverHook:
    def __init__(self, store_grad: bool = True) -> None:
        self.store_grad = store_grad
        self.grad_out = None

    def __call__(self, module, grad_input, grad_output) -> None:
        if self.store_grad:
            self.grad_out = grad_output[0]


class GetLayerGrad:
    """∂KL(softmax(out_q) ‖ softmax(out_fp))/∂(target output) with the network quantised up to and including the target
    (data_utill.py:190-264)."""

    def __init__(self, model: QuantModel, layer: Union[QuantLayer, BaseQuantBlock], device: torch.device,
                 use_aq: bool = False) -> None:
        self.model, self.layer, self.device, self.use_aq = model, layer, device, use_aq
        self.data_saver = GradSaverHook(True)

    def _quantize_model_till(self):
        self.model.set_quant_state(False, False)
        for m in self.model.modules():          # registration order == execution order for the UNets considered
            if isinstance(m, (QuantLayer, BaseQuantBlock)):
                m.set_quant_state(True, self.use_aq)
            if m is self.layer:
                break

    def __call__(self, xs: torch.Tensor, ts: torch.Tensor, cs: torch.Tensor = None) -> torch.Tensor:
        self.model.eval()
        handle = self.layer.register_full_backward_hook(self.data_saver)
        with torch.enable_grad():
            self.model.zero_grad()
            self.model.set_quant_state(False, False)
            out_fp = _run(self.model, xs, ts, cs, self.device)
            self._quantize_model_till()
            out_q = _run(self.model, xs, ts, cs, self.device)
            out_fp = out_fp[0] if isinstance(out_fp, (list, tuple)) else out_fp
            out_q = out_q[0] if isinstance(out_q, (list, tuple)) else out_q
            loss = F.kl_div(F.log_softmax(out_q, dim=1), F.softmax(out_fp, dim=1), reduction="batchmean")
            loss.backward()
        handle.remove()
        self.model.set_quant_state(False, False)
        self.layer.set_quant_state(True, self.use_aq)
        self.model.train()
        return self.data_saver.grad_out.data


def save_grad(model: QuantModel, layer: Union[QuantLayer, BaseQuantBlock], cali_data: Tuple[torch.Tensor], damping: float = 1.0,
              use_aq: bool = False, batch_size: int = 32, keep_gpu: bool = True) -> torch.Tensor:
    """|∂loss/∂output| + 1 of `layer` over the calibration set (data_utill.py:55-74; the reference's call sites pass
    ``asym`` in the ``damping`` position, which is unused there as well)."""
    device = next(model.parameters()).device
    get_grad = GetLayerGrad(model, layer, device, use_aq)
    grads = [get_grad(*(t[i: i + batch_size] for t in cali_data)).cpu() for i in range(0, cali_data[0].size(0), batch_size)]
    cached = torch.cat(grads).abs() + 1.0
    return cached.to(device) if keep_gpu else cached
