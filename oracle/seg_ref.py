"""oracle/seg_ref.py -- TEST INFRASTRUCTURE ONLY (checker, never the product).

CPU restatement, in plain PyTorch ops, of the steps either side of DGCNN.forward in the reference's loops
(SURVEY.md 8f rows N2-N4):

    step_loss     <- train.py:180-201 (training step: softmax, mask with the extra "ignored" class, criterion, arg-max)
                     and train.py:267-271 (validation step: the same without the mask)
    cal_loss      <- util.py:17-34     (label smoothing 0.2, log_softmax applied to its input, mean over the rows)
    predict_step  <- test.py:123-127, :308-311 (softmax, max, confidence threshold -> UNCLASSIFIED = 31)
    Adam          <- torch.optim.Adam itself, as constructed at train.py:118 (used directly by the tests)

The masked rows are built with tensor ops instead of the reference's per-point Python loops (train.py:186-191); the
rows are the same: (p, 0) with the point's label where the mask is set, (0, .., 0, 1) with label num_classes elsewhere.

Parity status: PINNED.  util.py / train.py / test.py cannot be imported in this image (prettytable, tensorboard, h5py,
laspy are absent -- SURVEY.md 8c caveat 2), so tests/golden/gen_golden.py takes the reference's own statements for
these steps out of the files with `ast` (util.py:17-34, train.py:180-197 + :201, :268-271, test.py:123-127), executes
them on fixed inputs and commits the results as tests/golden/seg_steps.npz; tests/test_oracle_golden.py holds this
module to them on every run.
Only tests/ and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F

UNCLASSIFIED = 31  # test.py: label given to points below the confidence threshold


def cal_loss(pred, gold, smoothing=True, eps=0.2):
    gold = gold.contiguous().view(-1)
    if not smoothing:
        return F.cross_entropy(pred, gold, reduction="mean")
    n_class = pred.size(1)
    one_hot = torch.zeros_like(pred).scatter(1, gold.view(-1, 1), 1)
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / (n_class - 1)
    log_prb = F.log_softmax(pred, dim=1)
    return -(one_hot * log_prb).sum(dim=1).mean()


def step_loss(logits, labels, mask=None, eps=0.2):
    """logits [B, num_classes, N], labels [B, N], mask [B, N] or None -> (loss, pred [B, N])."""
    B, C, N = logits.shape
    prob = F.softmax(logits, dim=1)
    labels = labels.long()
    if mask is not None:
        on = (mask != 0)
        rows = torch.cat((prob * on.unsqueeze(1), (~on).unsqueeze(1).to(prob.dtype)), dim=1)   # [B, C + 1, N]
        labels = torch.where(on, labels, torch.full_like(labels, C))
        C = C + 1
    else:
        rows = prob
    rows = rows.permute(0, 2, 1).contiguous()
    loss = cal_loss(rows.view(-1, C), labels.view(-1, 1).squeeze(), smoothing=True, eps=eps) if eps else \
        cal_loss(rows.view(-1, C), labels.view(-1), smoothing=False)
    return loss, rows.max(dim=2)[1]


def predict_step(logits, min_class_confidence, unclassified=UNCLASSIFIED):
    prob = F.softmax(logits.permute(0, 2, 1).contiguous(), dim=2)
    vals, pred = prob.max(dim=2)
    pred[torch.where(vals < min_class_confidence)] = unclassified
    return pred, vals
