"""The oracle (oracle/cgat_oracle.py) against vectors produced by the unmodified reference.

Tolerance: max-norm relative <= 1e-5 here (same fp32 CPU arithmetic, same op order -- the
oracle is expected to match far tighter than the 1e-4 bar the HIP path is held to)."""
import types

import pytest

import recipe
from golden_util import check_case
from oracle import cgat_oracle as O

NS = types.SimpleNamespace(
    MultiHeadNetwork=O.MultiHeadNetwork, GATConvNodes=O.GATConvNodes, GATConvEdges=O.GATConvEdges,
    MHAttention=O.MHAttention, CGAtNet=O.CGAtNet, H_Net_0=O.H_Net_0, H_Net=O.H_Net,
    SimpleNetwork=O.SimpleNetwork, ResidualNetwork=O.ResidualNetwork, WeightedAttention=O.WeightedAttention,
    MessageLayer=O.MessageLayer, Roost=O.Roost, RoostSimpleNetwork=O.SimpleNetwork)

TINY = recipe.tiny_cases(NS)
BASE = recipe.base_cases(NS)


@pytest.mark.parametrize("cname", sorted(TINY))
def test_oracle_tiny(cname):
    check_case("tiny.npz", cname, TINY[cname], tol=1e-5)


@pytest.mark.parametrize("cname", sorted(BASE))
def test_oracle_base(cname):
    check_case("base.npz", cname, BASE[cname], tol=1e-5)


def test_state_dict_layout_matches_survey():
    """Appendix B of SURVEY.md: 307 tensors, 44 593 247 parameters at (200,128,L=4,H=3)."""
    m = O.CGAtNet(200, 128, 4, msg_heads=3, update_edges=True)
    sd = m.state_dict()
    assert len(sd) == 307
    assert sum(p.numel() for p in m.parameters()) == 44593247
    assert tuple(sd["graphs.0.Node.MH_A.fc_in.weight"].shape) == (768, 384, 1)
    assert tuple(sd["graphs.0.Node.Pooling_NN.Hyper.layers.0.hyper_linear.hypo_params.net.4.weight"].shape) == (16512, 128)
    assert tuple(sd["graphs.1.Node.Pooling_NN.damping"].shape) == (1,)
    assert "graphs.0.Node.Pooling_NN.damping" not in sd
