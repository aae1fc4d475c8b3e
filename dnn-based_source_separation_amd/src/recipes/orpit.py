"""One-and-rest recursive separation: a Conv-TasNet trained with ORPIT has TWO outputs, "one
speaker" and "the rest"; an n-speaker mixture is separated by feeding the rest back into the same model n - 2 more times.
Counterpart in the reference: Tester.run of egs/wsj0-mix/orpit_conv-tasnet/src/adhoc_driver.py:185-207 (its BSS-eval and
PESQ post-processing call external tools and is not part of the separation path).  Every model call is the fused forward
of models.conv_tasnet.ConvTasNet; the final assignment is criterion.pit.PIT1d (one sep_sisdr_dots + sep_pit_search launch)."""
import torch


def separate_one_and_rest(model, mixture, n_sources):
    """mixture (batch_size, 1, T) -> estimates (batch_size, n_sources, T): [one_1, one_2, ..., one_{n-1}, rest_{n-1}]
    (adhoc_driver.py:193-204)."""
    if n_sources < 2:
        raise ValueError("n_sources is expected >= 2, but given {}".format(n_sources))
    outputs = []
    rest = mixture
    for _ in range(n_sources - 1):
        one_and_rest = model(rest)
        if one_and_rest.size(1) != 2:
            raise ValueError("an ORPIT model has 2 outputs (one, rest), this one has {}".format(one_and_rest.size(1)))
        one, rest = torch.split(one_and_rest, [1, 1], dim=1)
        outputs.append(one)
    outputs.append(rest)
    return torch.cat(outputs, dim=1)


class ORPITEvaluator:
    """Loss and loss improvement of the recursive separation per utterance, as the reference's Tester prints them
    (adhoc_driver.py:190-207): `pit_criterion` is a PIT1d over the utterance's n_sources."""

    def __init__(self, model, pit_criterion, n_sources):
        self.model, self.pit_criterion, self.n_sources = model, pit_criterion, n_sources

    @torch.no_grad()
    def __call__(self, mixture, sources):
        """mixture (B, 1, T), sources (B, n_sources, T) -> dict(output, loss (B,), loss_improvement (B,), perm_idx (B, n_sources))"""
        self.model.eval()
        loss_mixture, _ = self.pit_criterion(mixture.expand_as(sources).contiguous(), sources, batch_mean=False)
        output = separate_one_and_rest(self.model, mixture, self.n_sources)
        loss, perm_idx = self.pit_criterion(output, sources, batch_mean=False)
        return {"output": output, "loss": loss, "loss_improvement": loss_mixture - loss, "perm_idx": perm_idx}
