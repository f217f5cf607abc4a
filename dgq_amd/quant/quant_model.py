"""QuantModel — mirror of the reference's ``quant/quant_model.py:18-205``: recursively wraps every
nn.Linear / nn.Conv2d into a QuantLayer and every ResnetBlock2D / BasicTransformerBlock into its Quant block,
keeps the ``config`` shim the diffusers pipeline reads, the state switches, ``half/float`` and ``device``."""
from typing import List

import torch
import torch.nn as nn

from .adaptive_rounding import AdaRoundQuantizer
from .quant_block import BaseQuantBlock, QuantBasicTransformerBlock, QuantResnetBlock2D, b2qb
from .quant_layer import QMODE, QuantLayer, SlotRef, StraightThrough, UniformAffineQuantizer
from .quant_layer_text import T2ILogQuantizer


class CFG:
    in_channels = 0
    sample_size = 0
    time_cond_proj_dim = 0
    addition_time_embed_dim = 0


def _cfg_get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


class QuantModel(nn.Module):
    def __init__(self, model: nn.Module, wq_params: dict = {}, aq_params: dict = {}, softmax_aq_params: dict = {},
                 cali: bool = True, tib_recon: bool = False, **kwargs) -> None:
        super().__init__()
        if tib_recon:
            raise NotImplementedError("tib_recon is reconstruction-time (quant_model.py:52-64); inference uses False")
        self.model = model
        self.config = CFG()                                           # for the diffusers pipeline
        self.config.in_channels = _cfg_get(model.config, "in_channels")
        self.config.sample_size = _cfg_get(model.config, "sample_size")
        self.config.time_cond_proj_dim = _cfg_get(model.config, "time_cond_proj_dim")
        if _cfg_get(model.config, "addition_time_embed_dim") is not None:
            self.config.addition_time_embed_dim = _cfg_get(model.config, "addition_time_embed_dim")
        self.tib_recon = tib_recon
        self.B = b2qb()
        self.slot_ref = SlotRef()
        self.quant_module(self.model, wq_params, aq_params, aq_mode=kwargs.get("aq_mode", [QMODE.NORMAL.value]),
                          prev_name=None)
        self.quant_block(self.model, wq_params, aq_params, softmax_aq_params)
        from .quant_block import TembGroup, CtxGroup
        self._epoch = [0]               # one tick per run of the wrapped UNet: scopes the shared-launch caches below
        ctx_layers = [l for m in self.model.modules() if isinstance(m, QuantBasicTransformerBlock)
                      for l in (m.attn2.to_k, m.attn2.to_v) if isinstance(l, QuantLayer)]
        if ctx_layers:
            cgrp = CtxGroup(ctx_layers, self._epoch)
            for l in ctx_layers:
                l.__dict__["_ctx_group"] = cgrp
        temb_layers = [m.time_emb_proj for m in self.model.modules()
                       if isinstance(m, QuantResnetBlock2D) and isinstance(m.time_emb_proj, QuantLayer)]
        if temb_layers:
            grp = TembGroup(temb_layers, self._epoch)
            for l in temb_layers:
                l.__dict__["_temb_group"] = grp          # plain attribute: not a submodule, not in the state dict
        # the epoch also ticks when the wrapped UNet is run directly (qnn.model(...), a hook or test that re-enters it): a
        # forward pre-hook on the UNet itself, so the shared-launch caches can never serve a previous run's projections
        self.model.register_forward_pre_hook(lambda mod, args: self._tick())
        self.time_aware = None          # set by load_cali_model(time_aware_aqtizer=True)
        self._graphs = None
        self._graph_pool = None

    # -- module surgery (quant_model.py:66-103) ------------------------------------------------------------
    def quant_module(self, module: nn.Module, wq_params: dict = {}, aq_params: dict = {},
                     aq_mode: List[int] = [QMODE.NORMAL.value], prev_name: str = None) -> None:
        for name, child in module.named_children():
            if isinstance(child, tuple(QuantLayer.QMAP.keys())):
                ql = QuantLayer(child, wq_params, aq_params, aq_mode=aq_mode)
                ql._slot_ref = self.slot_ref
                setattr(module, name, ql)
            elif isinstance(child, StraightThrough):
                continue
            else:
                self.quant_module(child, wq_params, aq_params, aq_mode=aq_mode, prev_name=name)

    def quant_block(self, module: nn.Module, wq_params: dict = {}, aq_params: dict = {},
                    softmax_aq_params: dict = {}) -> None:
        for name, child in module.named_children():
            cls = self.B.get(child.__class__.__name__)
            if cls is QuantBasicTransformerBlock:
                setattr(module, name, cls(child, aq_params, softmax_aq_params))
            elif cls is QuantResnetBlock2D:
                setattr(module, name, cls(child, aq_params))
            else:
                self.quant_block(child, wq_params, aq_params, softmax_aq_params)

    # -- state switches ----------------------------------------------------------------------------------------
    def _drop_graphs(self):
        """Captured graphs bake in the quantisation state, the dtype and the buffers of the moment of capture: any change
        of those invalidates them (they are re-captured on the next forward)."""
        if self._graphs is not None:
            self._graphs = {}
            self._graph_pool = None

    def set_quant_state(self, use_wq: bool = False, use_aq: bool = False) -> None:
        self._drop_graphs()
        for m in self.model.modules():
            if isinstance(m, (BaseQuantBlock, QuantLayer)):
                m.set_quant_state(use_wq=use_wq, use_aq=use_aq)

    def disable_out_quantization(self) -> None:
        """conv_in / conv_out stay floating point (quant_model.py:118-124)."""
        self._drop_graphs()
        self.model.conv_in.use_wq = False
        self.model.conv_in.disable_aq = True
        self.model.conv_out.use_wq = False
        self.model.conv_out.disable_aq = True

    def forward(self, sample, timesteps, encoder_hidden_states, *args, **kwargs):
        slot = None
        if self.time_aware is not None:
            # time-aware activation tables (calibration.py:297-312): the reference re-copies ~750 δ/z tensors
            # host->device here; every slot is already device-resident, so this only flips an index.
            t = timesteps if not torch.is_tensor(timesteps) else (timesteps if timesteps.dim() == 0 else timesteps[0])
            n = self.time_aware["num_inference_steps"]
            slot = int((1000 - int(t)) // (1000 // n))
            self.activate_slot(slot)
        if (self._graphs is not None and torch.is_tensor(sample) and sample.is_cuda and not args
                and self._graphable_kwargs(kwargs)):
            return self._graph_forward(slot, sample, timesteps, encoder_hidden_states, kwargs)
        return self._run_model(sample, timesteps, encoder_hidden_states, *args, **kwargs)

    def _tick(self):
        self._epoch[0] += 1

    def _run_model(self, *args, **kwargs):
        try:
            return self.model(*args, **kwargs)           # the pre-hook ticks the epoch
        finally:
            self._release_shared()

    def _release_shared(self):
        """drop what the shared-launch caches pinned for this run (the context / temb tensors and every projection output)"""
        for m in self.model.modules():
            for g in (m.__dict__.get("_ctx_group"), m.__dict__.get("_temb_group")):
                if g is not None:
                    g._src, g._out = None, {}

    # -- hipGraph replay of a whole denoise step -------------------------------------------------------------------
    def enable_graphs(self, enabled: bool = True):
        """Capture one hipGraph per (timestep slot, input signature) and replay it afterwards: a step is ~1400 kernel
        launches, which eager Python issues slower than the GPU executes them.  Every kernel of this package takes
        its stream explicitly and neither allocates nor synchronises, so the whole forward is capturable."""
        self._graphs = {} if enabled else None
        self._graph_pool = None
        return self

    @staticmethod
    def _graphable_kwargs(kwargs):
        """Only the keyword arguments the diffusers pipelines pass with no effect on this UNet (None / False / empty) may
        accompany a replayed graph; anything else runs eagerly instead of being silently frozen at its first value."""
        for k, v in kwargs.items():
            if k == "added_cond_kwargs":
                if v is not None and not all(torch.is_tensor(t) for t in v.values()):
                    return False
            elif not (v is None or v is False or (isinstance(v, (dict, list, tuple)) and len(v) == 0)):
                return False
        return True

    def _graph_forward(self, slot, sample, timesteps, ehs, kwargs):
        ack = kwargs.get("added_cond_kwargs") or None
        extra = {k: v for k, v in kwargs.items() if k != "added_cond_kwargs"}
        traw = timesteps if not torch.is_tensor(timesteps) else (timesteps if timesteps.dim() == 0 else timesteps[0])
        t_is_float = (torch.is_tensor(traw) and traw.is_floating_point()) or isinstance(traw, float)
        tval = float(traw) if t_is_float else int(traw)
        key = (slot, tuple(sample.shape), sample.dtype, tuple(ehs.shape), ehs.dtype, t_is_float,
               tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in ack.items())) if ack else None)
        ent = self._graphs.get(key)
        if ent is None:
            dev = sample.device
            st = dict(sample=sample.clone(), ehs=ehs.clone(),
                      t=torch.full((1,), tval, dtype=torch.float32 if t_is_float else torch.int64, device=dev),
                      ack={k: v.clone() for k, v in ack.items()} if ack else None)
            kw = dict(extra)
            if st["ack"] is not None:
                kw["added_cond_kwargs"] = st["ack"]
            with torch.no_grad():
                self._run_model(st["sample"], st["t"], st["ehs"], **kw)     # eager warm-up: lazy inits, caches
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                if self._graph_pool is None:
                    self._graph_pool = torch.cuda.graph_pool_handle()
                with torch.cuda.graph(g, pool=self._graph_pool):
                    out = self._run_model(st["sample"], st["t"], st["ehs"], **kw)
            ent = (g, st, out)
            self._graphs[key] = ent
        g, st, out = ent
        st["sample"].copy_(sample)
        st["ehs"].copy_(ehs)
        st["t"].fill_(tval)
        if ack:
            for k, v in ack.items():
                st["ack"][k].copy_(v)
        g.replay()
        return [o.clone() for o in out]

    def activate_slot(self, slot: int):
        ta = self.time_aware
        if slot not in ta["slots"]:
            raise KeyError("cali_ckpt has no 'act_%d' (needs act_0..act_%d for %d inference steps)"
                           % (slot, ta["num_inference_steps"] - 1, ta["num_inference_steps"]))
        if self.slot_ref.slot == slot:
            return
        self.slot_ref.slot = slot
        for q, table in ta["attn"]:                 # attention-side quantizers: swap the (device) tensors
            d, z = table[slot]
            q.delta.data = d
            q.zero_point.data = z

    def prepare_slots(self, slots=None):
        """Plan + upload the activation tables (and per-slot permuted weight copies) of the given timestep
        slots ahead of time, so the first forward of a slot does no host work."""
        if self.time_aware is None:
            return
        slots = sorted(self.time_aware["slots"]) if slots is None else slots
        prev = self.slot_ref.slot
        for s in slots:
            self.slot_ref.slot = s
            for m in self.model.modules():
                if isinstance(m, QuantLayer) and m.use_wq and m.use_aq and not m.disable_aq and s in m._act_tables:
                    m._binding()
        self.slot_ref.slot = prev

    # -- calibration-time API (producer side; quant_model.py:135-181) ---------------------------------
    def _attn_quantizers(self, with_w=False):
        for m in self.model.modules():
            if isinstance(m, QuantBasicTransformerBlock):
                for attn in (m.attn1, m.attn2):
                    for n in ("aqtizer_q", "aqtizer_k", "aqtizer_v") + (("aqtizer_w",) if with_w else ()):
                        yield getattr(attn, n)

    def set_group_num(self, group_num: int = 1) -> None:
        """Start DGQ statistics collection: every following forward records per-axis min / max in each activation
        quantizer (QuantLayer inputs, unfolded for convs; attention q / k / v).  The blocks run unfused meanwhile."""
        from . import quant_block
        self._drop_graphs()
        if not getattr(self, "_calibrating", False):      # a second set_group_num before done_group_num keeps the saved flag
            self._fusion_was = quant_block.FUSION
            self._calibrating = True
        quant_block.FUSION = False
        for m in self.model.modules():
            if isinstance(m, QuantLayer):
                m.set_group_num(group_num)
        for q in self._attn_quantizers():
            q.group_num = group_num

    def done_group_num(self, group_num, mode) -> None:
        """Turn the recorded statistics into grouped (δ, z) tables (K-Means on the host) and leave calibration mode."""
        from . import quant_block
        for m in self.model.modules():
            if isinstance(m, QuantLayer):
                m.done_group_num(group_num, mode=mode)
        for q in self._attn_quantizers():
            q.done_group_num(group_num, mode=mode)
        self.restore_fusion()
        self._drop_graphs()

    def restore_fusion(self):
        """undo set_group_num's switch to the unfused graph (also called by cali_model_aq when a calibration forward raises)"""
        from . import quant_block
        if getattr(self, "_calibrating", False):
            quant_block.FUSION = getattr(self, "_fusion_was", True)
            self._calibrating = False

    def set_running_stat(self, running_stat: bool = False) -> None:
        self._drop_graphs()
        for q in self._attn_quantizers(with_w=True):
            q.running_stat = running_stat
        for m in self.model.modules():
            if isinstance(m, QuantLayer):
                m.set_running_stat(running_stat)

    def synchorize_activation_statistics(self):
        raise NotImplementedError("multi-GPU calibration is disabled in the reference too (src/quantize_weight.py:214)")

    # -- dtype / device ---------------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        self._drop_graphs()                       # .to() / .cuda() / .half() / .float() move or recast the captured buffers
        return super()._apply(fn, *a, **k)

    def half(self):
        super().half()
        for m in self.model.modules():
            if isinstance(m, (AdaRoundQuantizer, UniformAffineQuantizer, QuantLayer)):
                m.half()
        return self

    def float(self):
        super().float()
        for m in self.model.modules():
            if isinstance(m, (AdaRoundQuantizer, UniformAffineQuantizer, QuantLayer)):
                m.float()
        return self

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype
