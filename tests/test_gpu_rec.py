"""GPU parity of the whole libs/lasagne_libs/recurrent_layers.py family (20 classes)
against the oracle's variant-table layer (itself pinned class by class by torch autograd in
tests/test_oracle_autograd.py): 2-layer bidirectional nets so that the own-input conditioning
gradient of layer 2 flows into layer 1, ragged batch, active grad_clip."""
import numpy as np
import pytest

from asr_amd import specs
from common import make_batch
from test_gpu_parity import perturb_params, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('cls', sorted(specs.REC_CLASSES))
def test_recurrent_layers_family_small(cls):
    from asr_amd.gating_hyper_nets import _build_rec
    v = specs.REC_CLASSES[cls]
    I = 3 if v['cond'] == 'ext' else 0
    net, cond_layers = _build_rec(5, 6, 2, 8, 0.05, 0.0, cls, cls if v['cond'] else None, 2, num_factors=3 if v['cond'] else 4,
                                  num_prjs=4, cond_dim=I, device='cuda:0', seed=3)
    assert len(cond_layers) == (4 if v['cond'] else 0)
    perturb_params(net, 21, 0.15)
    b = make_batch(np.random.default_rng(22), 3, 7, 5, 6, I=I)
    run_case(net, b, cond_key='ivec' if I else None)
    if v['cond']:
        assert cond_layers[0].get_sample_feat().shape == (3, 3)


@pytest.mark.parametrize('cls', ['CondLayerNormProjectLSTM_v0_1_Layer', 'MaxCondLayerNormProjectLSTMLayer',
                                 'LayerNormLSTMLayer', 'CondLayerNormLSTMLayer'])
def test_recurrent_layers_family_medium(cls):
    """N=64 (16 workgroups per direction), B=18 > 16 (two batch chunks), T=29."""
    from asr_amd.gating_hyper_nets import _build_rec
    v = specs.REC_CLASSES[cls]
    net, _ = _build_rec(11, 13, 2, 64, 1.0, 0.0, cls, cls if v['cond'] else None, 2, num_factors=8 if v['cond'] else 32,
                        num_prjs=32, device='cuda:0', seed=4)
    perturb_params(net, 23, 0.03)
    b = make_batch(np.random.default_rng(24), 18, 29, 11, 13)
    run_case(net, b, calibrate=True)
