"""Mirror of the reference's ``quant`` package surface for the inference path and the
calibration producers of §8(f): activation grouping (act_group_quant) and weight PTQ (cali_model)."""
from .quant_layer import (Scaler, QMODE, StraightThrough, UniformAffineQuantizer, QuantLayer, minmax)
from .quant_layer_text import T2ILogQuantizer
from .adaptive_rounding import AdaRoundQuantizer, RMODE
from .quant_block import BaseQuantBlock, QuantResnetBlock2D, QuantBasicTransformerBlock, b2qb
from .quant_model import QuantModel
from .calibration import load_cali_model, cali_model
from .load_qmodel_util import get_qmodel
from .calibration_group_quantization import act_group_quant, cali_model_aq
from .reconstruction import layer_reconstruction, block_reconstruction
from .reconstruction_util import RLOSS, LossFunc
