import torch
import torch.nn as nn
import torch.nn.functional as F

from gnf_hip import ops


def _pairs(seq):
    return [(m.weight, m.bias) for m in seq if isinstance(m, nn.Linear)]


class MLP(nn.Module):
    """Linear/act chain (reference models/MLP.py:6-21); ReLU chains run on the MFMA GEMM."""

    def __init__(self, in_d, hidden, out_d, act_f=nn.ReLU()):
        super().__init__()
        self.in_d = in_d
        self.hiddens = hidden
        self.out_d = out_d
        self.act_f = act_f
        layers_dim = [in_d] + hidden + [out_d]
        layers = []
        for dim_in, dim_out in zip(layers_dim[:-1], layers_dim[1:]):
            layers += [nn.Linear(dim_in, dim_out), act_f]
        layers.pop()
        self.net = nn.Sequential(*layers)

    def forward(self, x, context=None):
        if isinstance(self.act_f, nn.ReLU):
            return ops.mlp(x, _pairs(self.net))
        return self.net(x)


def _conv3x3(x, weight, bias):
    """valid 3x3 convolution as im2col (F.unfold) + one batched matmul; same values as nn.Conv2d."""
    n, c, h, w = x.shape
    cols = F.unfold(x, 3)                                        # [n, c*9, (h-2)*(w-2)]
    y = torch.matmul(weight.view(weight.shape[0], -1), cols) + bias.view(1, -1, 1)
    return y.view(n, weight.shape[0], h - 2, w - 2)


class MNISTCNN(nn.Module):
    """Embedding net of the MNIST DAG flow (reference models/MLP.py:24-48): conv3x3(1->16) ReLU
    conv3x3(16->16) maxpool2 flatten fc(2304->128) ReLU fc(128->out_d).  For the 28x28
    single-channel case the convolutional front is one fused LDS-resident MFMA kernel
    (gnf_mnistcnn.hip) and the fc layers run on the MFMA GEMM chain."""

    def __init__(self, out_d=10, fc_l=[2304, 128], size_img=[1, 28, 28]):
        super(MNISTCNN, self).__init__()
        self.conv1 = nn.Conv2d(size_img[0], 16, 3, 1)
        self.conv2 = nn.Conv2d(16, 16, 3, 1)
        self.dropout1 = nn.Dropout2d(0.25)      # unused, as in the reference (:42,46)
        self.dropout2 = nn.Dropout2d(0.5)
        self.fc1 = nn.Linear(fc_l[0], fc_l[1])
        self.fc2 = nn.Linear(fc_l[1], out_d)
        self.out_d = out_d
        self.size_img = size_img

    def _fused_conv_ok(self, x):
        return (x.is_cuda and list(self.size_img) == [1, 28, 28] and x.shape[-1] == 784
                and tuple(self.conv1.weight.shape) == (16, 1, 3, 3) and tuple(self.conv2.weight.shape) == (16, 16, 3, 3))

    def forward(self, x, context=None):
        b_size = x.shape[0]
        if self._fused_conv_ok(x):
            x = ops.MnistConvFn.apply(x.view(-1, 784), self.conv1.weight, self.conv1.bias, self.conv2.weight,
                                      self.conv2.bias)
        else:   # other image sizes (the 14x14 / 7x7 scales of the multi-scale factory, 7 % of its images):
            # im2col + library GEMM (no MIOpen: its find step costs minutes on this stack)
            x = _conv3x3(x.view(-1, self.size_img[0], self.size_img[1], self.size_img[2]), self.conv1.weight,
                         self.conv1.bias)
            x = F.relu(x)
            x = _conv3x3(x, self.conv2.weight, self.conv2.bias)
            x = F.max_pool2d(x, 2)
            x = torch.flatten(x, 1)
        x = ops.mlp(x, [(self.fc1.weight, self.fc1.bias), (self.fc2.weight, self.fc2.bias)])
        return x.view(b_size, -1)


class CIFAR10CNN(nn.Module):
    """reference models/MLP.py:51-72; not on any measured configuration (plain torch)."""

    def __init__(self, out_d=10, fc_l=[400, 128, 84], size_img=[3, 32, 32], k_size=5):
        super(CIFAR10CNN, self).__init__()
        self.conv1 = nn.Conv2d(size_img[0], 6, k_size)
        self.pool = nn.MaxPool2d(2, 2)
        self.conv2 = nn.Conv2d(6, 16, k_size)
        self.fc1 = nn.Linear(fc_l[0], fc_l[1])
        self.fc2 = nn.Linear(fc_l[1], fc_l[2])
        self.fc3 = nn.Linear(fc_l[2], out_d)
        self.out_d = out_d
        self.size_img = size_img

    def forward(self, x, context=None):
        b_size = x.shape[0]
        x = self.pool(F.relu(self.conv1(x.view(-1, self.size_img[0], self.size_img[1], self.size_img[2]))))
        x = self.pool(F.relu(self.conv2(x)))
        x = x.view(b_size, -1)
        x = ops.mlp(x, [(self.fc1.weight, self.fc1.bias), (self.fc2.weight, self.fc2.bias),
                        (self.fc3.weight, self.fc3.bias)])
        return x.view(b_size, -1)


class IdentityNN(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x, context=None):
        return x
