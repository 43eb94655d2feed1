"""The policy/value network that sits on the other side of the `actor` batch (cfg 5).

PyTorch-ROCm plumbing, not part of the hot path: the shape of the reference's `Model_PolicyValue`
(go/df_model2.py:28-94: a 3x3 conv + BatchNorm + LeakyReLU(0.1) stem, `num_block` residual blocks of two
such layers at `dim` channels, 1x1 heads, softmax policy over the 361 points, tanh value) with the
reference's parameter names (`init_conv`, `conv_lower<i>`, `conv_upper<i>`, `pi_final_conv`,
`value_final_conv`, `pi_linear`, `value_linear1`, `value_linear2`), so a state dict saved by the
reference loads here.  Random-initialised when no checkpoint is given (none exists in this repo).
"""
import torch
import torch.nn as nn


class PolicyValueNet(nn.Module):
    def __init__(self, num_planes=25, dim=128, num_block=20, board_size=19, leaky_relu=True, bn=True):
        super().__init__()
        self.board_size, self.dim, self.num_block, self.bn = board_size, dim, num_block, bn
        self.relu = nn.LeakyReLU(0.1) if leaky_relu else nn.ReLU()
        self.init_conv = self._layer(num_planes, dim)
        for i in range(num_block):
            setattr(self, "conv_lower%d" % i, self._layer(dim, dim))
            setattr(self, "conv_upper%d" % i, self._layer(dim, dim, relu=False))
        self.pi_final_conv = self._layer(dim, 2, kernel=1)
        self.value_final_conv = self._layer(dim, 1, kernel=1)
        d = board_size * board_size
        self.pi_linear = nn.Linear(2 * d, d)
        self.value_linear1 = nn.Linear(d, 256)
        self.value_linear2 = nn.Linear(256, 1)

    def _layer(self, cin, cout, kernel=3, relu=True):
        mods = [nn.Conv2d(cin, cout, kernel, padding=kernel // 2)]
        if self.bn:
            mods.append(nn.BatchNorm2d(cout))
        if relu:
            mods.append(self.relu)
        return nn.Sequential(*mods)

    def forward(self, x):
        """x: the batch (anything answering ["s"]) or the planes tensor [B, 25, 19, 19]."""
        s = x if torch.is_tensor(x) else x["s"]
        s = self.init_conv(s.to(self.pi_linear.weight.dtype))
        for i in range(self.num_block):
            t = getattr(self, "conv_upper%d" % i)(getattr(self, "conv_lower%d" % i)(s))
            s = self.relu(t + s)
        d = self.board_size * self.board_size
        pi = torch.softmax(self.pi_linear(self.pi_final_conv(s).reshape(-1, 2 * d)), dim=1)
        v = self.relu(self.value_linear1(self.value_final_conv(s).reshape(-1, d)))
        return dict(pi=pi, V=torch.tanh(self.value_linear2(v)))


def make_actor(net):
    """The callback of the `actor` collector: what Evaluator.actor replies without a sampler
    (rlpytorch/trainer/trainer.py:57-91: pi, and V) -- the environment samples the move from pi over
    the legal points on the device."""
    def actor(batch):
        with torch.no_grad():
            out = net(batch.hist(0))
        return dict(pi=out["pi"].float(), V=out["V"].float().view(-1))
    return actor
