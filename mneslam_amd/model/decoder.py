"""Decoder parameter containers with the reference's module tree and state_dict keys
(model/decoder.py): ``color_net.model.{0,2}.weight``, ``sdf_net.model.{0,2}.weight``, bias-free
``nn.Linear`` with default init.  The hot path never calls these modules -- the HIP kernels read the
``.weight`` tensors directly; ``forward`` exists for API completeness (cold callers).
``decoder.tcnn_network`` (tinycudann FullyFusedMLP) is disabled in every shipped config and is not
provided."""
import torch
import torch.nn as nn


def _mlp(in_dim, hidden, out_dim, num_layers):
    layers = []
    for l in range(num_layers):
        i = in_dim if l == 0 else hidden
        o = out_dim if l == num_layers - 1 else hidden
        layers.append(nn.Linear(i, o, bias=False))
        if l != num_layers - 1:
            layers.append(nn.ReLU(inplace=True))
    return nn.Sequential(*layers)


class ColorNet(nn.Module):
    """reference: model/decoder.py:7-55"""

    def __init__(self, config, input_ch=4, geo_feat_dim=15, hidden_dim_color=64, num_layers_color=3):
        super().__init__()
        if config["decoder"]["tcnn_network"]:
            raise NotImplementedError("decoder.tcnn_network=True (tinycudann) is not part of this build")
        self.config = config
        self.input_ch, self.geo_feat_dim = input_ch, geo_feat_dim
        self.hidden_dim_color, self.num_layers_color = hidden_dim_color, num_layers_color
        self.model = _mlp(input_ch + geo_feat_dim, hidden_dim_color, 3, num_layers_color)

    def forward(self, input_feat):
        return self.model(input_feat)


class SDFNet(nn.Module):
    """reference: model/decoder.py:57-108"""

    def __init__(self, config, input_ch=3, geo_feat_dim=15, hidden_dim=64, num_layers=2):
        super().__init__()
        if config["decoder"]["tcnn_network"]:
            raise NotImplementedError("decoder.tcnn_network=True (tinycudann) is not part of this build")
        self.config = config
        self.input_ch, self.geo_feat_dim = input_ch, geo_feat_dim
        self.hidden_dim, self.num_layers = hidden_dim, num_layers
        self.model = _mlp(input_ch, hidden_dim, 1 + geo_feat_dim, num_layers)

    def forward(self, x, return_geo=True):
        out = self.model(x)
        return out if return_geo else out[..., :1]


class _ColorSDFBase(nn.Module):
    def __init__(self, config, color_in, sdf_in):
        super().__init__()
        dec = config["decoder"]
        self.config = config
        self.color_net = ColorNet(config, input_ch=color_in, geo_feat_dim=dec["geo_feat_dim"],
                                  hidden_dim_color=dec["hidden_dim_color"], num_layers_color=dec["num_layers_color"])
        self.sdf_net = SDFNet(config, input_ch=sdf_in, geo_feat_dim=dec["geo_feat_dim"],
                              hidden_dim=dec["hidden_dim"], num_layers=dec["num_layers"])

    def hip_weights(self):
        """(w_sdf0, w_sdf1, w_col0, w_col1) for the 2-layer/2-layer decoders the kernels support."""
        if self.sdf_net.num_layers != 2 or self.color_net.num_layers_color != 2:
            raise NotImplementedError("the HIP path supports decoder.num_layers == num_layers_color == 2 "
                                      "(every shipped config)")
        return (self.sdf_net.model[0].weight, self.sdf_net.model[2].weight,
                self.color_net.model[0].weight, self.color_net.model[2].weight)


class ColorSDFNet(_ColorSDFBase):
    """Colour planes + SDF planes (reference: model/decoder.py:110-141)."""

    def __init__(self, config, input_ch=3, input_ch_pos=12):
        super().__init__(config, color_in=input_ch + input_ch_pos, sdf_in=input_ch + input_ch_pos)

    def forward(self, embed, embed_pos, embed_color):
        h = self.sdf_net(torch.cat([embed, embed_pos], dim=-1), return_geo=True)
        sdf, geo = h[..., :1], h[..., 1:]
        rgb = self.color_net(torch.cat([embed_pos, embed_color, geo], dim=-1))
        return torch.cat([rgb, sdf], -1)


class ColorSDFNet_v2(_ColorSDFBase):
    """No colour planes (reference: model/decoder.py:143-175)."""

    def __init__(self, config, input_ch=3, input_ch_pos=12):
        super().__init__(config, color_in=input_ch_pos, sdf_in=input_ch + input_ch_pos)

    def forward(self, embed, embed_pos):
        h = self.sdf_net(torch.cat([embed, embed_pos], dim=-1), return_geo=True)
        sdf, geo = h[..., :1], h[..., 1:]
        rgb = self.color_net(torch.cat([embed_pos, geo], dim=-1))
        return torch.cat([rgb, sdf], -1)
