"""H_Net_0 / H_Net: hypernetwork "Pooling_NN" with the reference's parameter tree
(reference CGAT/Hypernetworksmp.py:24-313):

    Hyper.layers.{l}.hyper_linear.hypo_params.net.{s}.net.0.{weight,bias}   s < n_fc  (Linear+Tanh)
    Hyper.layers.{l}.hyper_linear.hypo_params.net.{n_fc}.{weight,bias}      Linear(W -> W*W+W)
    Hyper.layers.{last}.hypo_params. ...                                      (outermost HyperLinear)
    damping                                                                   (H_Net only)

The modules below only *hold* those parameters (constructed and initialised in the reference's
order, so torch.manual_seed gives the same initial values); the arithmetic is one call into
cgat_hnet_forward / cgat_hnet_backward, which never materialises the predicted [rows, W*W+W]
weights.
"""
import torch
import torch.nn as nn

from .ops import HNetFn


class FCLayer(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(in_features, out_features), nn.Tanh())


def _init_weights(m):
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight, a=0.0, nonlinearity='leaky_relu', mode='fan_in')


def _last_hyper_layer_init(m):
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight, a=0.0, nonlinearity='leaky_relu', mode='fan_in')
        m.weight.data *= 1e-1


class FCBlock(nn.Module):
    """Parameter holder of the hypernetwork trunk + head (Hypernetworksmp.py:36-83)."""

    def __init__(self, hidden_ch, num_hidden_layers, in_features, out_features, outermost_linear=False):
        super().__init__()
        net = [FCLayer(in_features, hidden_ch)]
        for _ in range(num_hidden_layers):
            net.append(FCLayer(hidden_ch, hidden_ch))
        net.append(nn.Linear(hidden_ch, out_features) if outermost_linear else FCLayer(hidden_ch, out_features))
        self.net = nn.Sequential(*net)
        self.net.apply(_init_weights)

    def __getitem__(self, item):
        return self.net[item]


class HyperLinear(nn.Module):
    def __init__(self, in_ch, out_ch, hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch):
        super().__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        self.hypo_params = FCBlock(in_features=hyper_in_ch, hidden_ch=hyper_hidden_ch,
                                   num_hidden_layers=hyper_num_hidden_layers,
                                   out_features=(in_ch * out_ch) + out_ch, outermost_linear=True)
        self.hypo_params[-1].apply(_last_hyper_layer_init)

    def flat_params(self):
        net = self.hypo_params.net
        trunk = [net[s].net[0] for s in range(len(net) - 1)]
        return [l.weight for l in trunk] + [l.bias for l in trunk] + [net[-1].weight, net[-1].bias]


class HyperLayer(nn.Module):
    def __init__(self, in_ch, out_ch, hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch):
        super().__init__()
        self.hyper_linear = HyperLinear(in_ch, out_ch, hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch)
        self.norm_nl = nn.Sequential(nn.LayerNorm([out_ch], elementwise_affine=False), nn.Tanh())


class HyperFC(nn.Module):
    def __init__(self, hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch, hidden_ch, num_hidden_layers, in_ch,
                 out_ch, outermost_linear=False):
        super().__init__()
        if not outermost_linear:
            raise NotImplementedError("the reference only ever builds HyperFC(outermost_linear=True) "
                                      "(Hypernetworksmp.py:274,305)")
        widths = {hyper_in_ch, hyper_hidden_ch, hidden_ch, in_ch, out_ch}
        if len(widths) != 1:
            raise NotImplementedError(f"HIP hypernetwork needs one common width, got {sorted(widths)} "
                                      "(CGAT.py:301-305 always passes equal widths)")
        self.width = in_ch
        self.n_fc = 1 + hyper_num_hidden_layers
        hk = dict(hyper_in_ch=hyper_in_ch, hyper_num_hidden_layers=hyper_num_hidden_layers,
                  hyper_hidden_ch=hyper_hidden_ch)
        self.layers = nn.ModuleList()
        self.layers.append(HyperLayer(in_ch=in_ch, out_ch=hidden_ch, **hk))
        for _ in range(num_hidden_layers):
            self.layers.append(HyperLayer(in_ch=hidden_ch, out_ch=hidden_ch, **hk))
        self.layers.append(HyperLinear(in_ch=hidden_ch, out_ch=out_ch, **hk))

    def run(self, h0, v, damping):
        flat = []
        for layer in self.layers:
            hl = layer.hyper_linear if isinstance(layer, HyperLayer) else layer
            flat += hl.flat_params()
        return HNetFn.apply(h0, v, damping, self.n_fc, len(self.layers), *flat)


class H_Net_0(nn.Module):
    """NN = Hyper(h_0); NN(x)   (Hypernetworksmp.py:257-285)."""

    def __init__(self, hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch, hidden_ch, num_hidden_layers, in_ch,
                 out_ch, outermost_linear=True):
        super().__init__()
        self.Hyper = HyperFC(hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch, hidden_ch, num_hidden_layers,
                             in_ch, out_ch, outermost_linear=True)
        self.out_ch = out_ch

    def forward(self, h_0, x):
        return self.Hyper.run(h_0, x, None)


class H_Net(nn.Module):
    """damping clamped to [0,1] in place on every forward; hyper input = d*h_0 + (1-d)*x; h_t is
    accepted and ignored, as in the reference (Hypernetworksmp.py:288-313)."""

    def __init__(self, hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch, hidden_ch, num_hidden_layers, in_ch,
                 out_ch, outermost_linear=True):
        super().__init__()
        self.Hyper = HyperFC(hyper_in_ch, hyper_num_hidden_layers, hyper_hidden_ch, hidden_ch, num_hidden_layers,
                             in_ch, out_ch, outermost_linear=True)
        self.damping = nn.Parameter(torch.rand(1))
        self.out_ch = out_ch

    def forward(self, h_0, h_t, x):
        with torch.no_grad():
            self.damping.data = self.damping.data.clamp(0.0, 1.0)
        return self.Hyper.run(h_0, x, self.damping)
