import torch
import torch.nn as nn

from gnf_hip import ops
from .Conditionners import Conditioner, DAGConditioner
from .Normalizers import Normalizer


class NormalizingFlow(nn.Module):
    """Abstract flow (reference models/NormalizingFlow.py:7-58)."""

    def __init__(self):
        super(NormalizingFlow, self).__init__()

    def forward(self, x, context=None):
        """-> (z, log|det J|)"""
        pass

    def constraintsLoss(self):
        pass

    def DAGness(self):
        pass

    def step(self, epoch_number, loss_avg):
        pass

    def getConditioners(self):
        pass

    def isInvertible(self):
        pass

    def getNormalizers(self):
        pass

    def invert(self, z, context=None):
        pass


class NormalizingFlowStep(NormalizingFlow):
    """h = conditioner(x); z, jac = normalizer(x, h); log|det J| = sum_i log jac
    (reference :61-107).  Normalizers exposing `forward_logdet` get the reduction fused into
    their kernel; any other Normalizer plug-in goes through the row-reduction kernel."""

    def __init__(self, conditioner: Conditioner, normalizer: Normalizer):
        super(NormalizingFlowStep, self).__init__()
        self.conditioner = conditioner
        self.normalizer = normalizer
        self.level_schedule = True       # invert(): topological level schedule for DAG conditioners

    def forward(self, x, context=None):
        h = self.conditioner(x, context)
        if hasattr(self.normalizer, "forward_logdet"):
            return self.normalizer.forward_logdet(x, h, context)
        z, jac = self.normalizer(x, h, context)
        return z, ops.LogSumRowsFn.apply(jac)

    def constraintsLoss(self):
        if type(self.conditioner) is DAGConditioner:
            return self.conditioner.loss()
        return 0.

    def DAGness(self):
        if type(self.conditioner) is DAGConditioner:
            return [self.conditioner.get_power_trace()]
        return [0.]

    def step(self, epoch_number, loss_avg):
        if type(self.conditioner) is DAGConditioner:
            self.conditioner.step(epoch_number, loss_avg)

    def getConditioners(self):
        return [self.conditioner]

    def getNormalizers(self):
        return [self.normalizer]

    def isInvertible(self):
        for conditioner in self.getConditioners():
            if not conditioner.is_invertible:
                return False
        return True

    def invert(self, z, context=None):
        """Fixed-point inverse: depth()+1 passes, early exit on exact equality (:98-107;
        the reference's progress print is dropped)."""
        x = torch.zeros_like(z)
        if type(self.conditioner) is DAGConditioner and context is None and self.level_schedule:
            # DAG-ordered inversion: every variable is inverted once, after its parents -- d conditioner rows in total
            # instead of (depth + 1) * d; same values as the fixed-point passes below (non-parents are masked by
            # exact zeros either way)
            P = self.conditioner.deterministic_importance()
            levels = self.conditioner.levels(P) if P is not None else None
            if levels is not None:
                with torch.no_grad():
                    for rows in levels:
                        h = self.conditioner.forward_rows(x, rows, P)
                        x[:, rows] = self.normalizer.inverse_transform(z[:, rows].contiguous(), h, context)
                return x
        with torch.no_grad():
            for i in range(self.conditioner.depth() + 1):
                h = self.conditioner(x, context)
                x_prev = x
                x = self.normalizer.inverse_transform(z, h, context)
                if torch.norm(x - x_prev) == 0.:
                    break
        return x


class FCNormalizingFlow(NormalizingFlow):
    """Stack of steps with the feature order reversed between steps (reference :110-169)."""

    def __init__(self, steps, z_log_density):
        super(FCNormalizingFlow, self).__init__()
        self.steps = nn.ModuleList()
        self.z_log_density = z_log_density
        for step in steps:
            self.steps.append(step)

    def forward(self, x, context=None):
        jac_tot = 0.
        for i, step in enumerate(self.steps):
            z, jac = step(x, context)
            if i + 1 < len(self.steps):
                x = torch.flip(z, dims=[1])      # z[:, inv_idx] of the reference (:120-123)
            jac_tot = jac_tot + jac
        return z, jac_tot                        # last step's z is returned un-flipped (:126)

    def constraintsLoss(self):
        loss = 0.
        for step in self.steps:
            loss += step.constraintsLoss()
        return loss

    def DAGness(self):
        dagness = []
        for step in self.steps:
            dagness += step.DAGness()
        return dagness

    def step(self, epoch_number, loss_avg):
        for step in self.steps:
            step.step(epoch_number, loss_avg)

    def loss(self, z, jac):
        log_p_x = jac + self.z_log_density(z)
        return self.constraintsLoss() - log_p_x.mean()

    def getNormalizers(self):
        normalizers = []
        for step in self.steps:
            normalizers += step.getNormalizers()
        return normalizers

    def getConditioners(self):
        conditioners = []
        for step in self.steps:
            conditioners += step.getConditioners()
        return conditioners

    def isInvertible(self):
        for conditioner in self.getConditioners():
            if not conditioner.is_invertible:
                return False
        return True

    def invert(self, z, context=None):
        """Exact inverse of forward for any number of steps.  The reference (:166-169) visits
        steps[-0] == steps[0] first and never undoes the flip, so it only inverts nb_flow=1
        flows; for nb_flow=1 this is identical to it."""
        n = len(self.steps)
        for s in range(n - 1, -1, -1):
            x = self.steps[s].invert(z, context)
            z = torch.flip(x, dims=[1]) if s > 0 else x
        return z


class CNNormalizingFlow(FCNormalizingFlow):
    """Multi-scale flow (reference :172-226): after every scale the image is cut into d_c x d_h x d_w blocks; the first
    element of each block goes on to the next (coarser) flow, the others are emitted as latent variables."""

    def __init__(self, steps, z_log_density, dropping_factors):
        super(CNNormalizingFlow, self).__init__(steps, z_log_density)
        self.dropping_factors = dropping_factors

    @staticmethod
    def _blocks(z, img_size, drop):
        """[B, C*H*W] -> [B, c, h, w, d_c*d_h*d_w]  (the unfold chain of reference :184-185)"""
        C, H, W = img_size
        d_c, d_h, d_w = drop
        c, h, w = int(C / d_c), int(H / d_h), int(W / d_w)
        return z.view(-1, c, d_c, h, d_h, w, d_w).permute(0, 1, 3, 5, 2, 4, 6).reshape(z.shape[0], c, h, w, -1)

    def forward(self, x, context=None):
        b_size = x.shape[0]
        jac_tot = 0.
        z_all = []
        for step, drop in zip(self.steps, self.dropping_factors):
            z, jac = step(x, context)
            blocks = self._blocks(z, step.img_sizes, drop)
            z_all.append(blocks[..., 1:].reshape(b_size, -1))
            x = blocks[..., 0].reshape(b_size, -1)
            jac_tot = jac_tot + jac
        z_all.append(x)
        return torch.cat(z_all, 1), jac_tot

    def invert(self, z, context=None):
        b_size = z.shape[0]
        parts, i = [], 0
        for step, drop in zip(self.steps, self.dropping_factors):
            C, H, W = step.img_sizes
            c, h, w = int(C / drop[0]), int(H / drop[1]), int(W / drop[2])
            nb_z = C * H * W - c * h * w if C * H * W != c * h * w else c * h * w
            parts.append(z[:, i:i + nb_z])
            i += nb_z
        x = None
        for k in range(len(self.steps) - 1, -1, -1):
            step, drop = self.steps[k], self.dropping_factors[k]
            C, H, W = step.img_sizes
            d_c, d_h, d_w = drop
            c, h, w = int(C / d_c), int(H / d_h), int(W / d_w)
            zk = parts[k]
            if c * h * w != C * H * W:       # re-interleave the kept element with the dropped ones of each block
                blocks = torch.cat((x.view(b_size, c, h, w, 1), zk.reshape(b_size, c, h, w, -1)), 4)
                zk = blocks.view(b_size, c, h, w, d_c, d_h, d_w).permute(0, 1, 4, 2, 5, 3, 6).reshape(b_size, -1)
            x = step.invert(zk.reshape(b_size, -1), context)
        return x

