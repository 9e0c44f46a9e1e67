"""Trainer-equivalent entry point: the behaviour of the reference's
trainer/train_{deepconn_pp,narre,dual_att}.py on top of the HIP modules (plus `--model simple_siamese` on the review
split: plain Adam, none of train_simple_siamese.py's SparseAdam / scheduler / review sampling).

    python -m review_based_recommender_amd.trainer --model deepconn --config cfg.json
    python -m torch.distributed.run --nproc-per-node 8 -m review_based_recommender_amd.trainer --model deepconn --config cfg.json

Kept from the reference: the flat JSON config keys (models/*/default_*.json), output directory
`./{log_dir}/{dataset}/{model_name}/{uid}` with `log.txt` (experiment.py:64-83), the args / parameter
tables (experiment.py:101-123), the step (train_deepconn_pp.py:161-168), the log line
`epoch: e/E, step: s/S, loss: ..., rmse: ..., lr: ..., gnorm: ..., time: ...` every `log_idx` steps
(:176-184), validation RMSE with best-model checkpoint `{"model","optimizer","updates","args"}`
(experiment.py:127-139, train_deepconn_pp.py:214-217) and patience early stop by raising `EarlyStop`
(:226-232).  Deliberate differences: `kernel_sizes` / `hidden_dim` come from the config unless
`--reference-quirks` restores the trainers' hard-coded `[3]` / 150 (train_deepconn_pp.py:125,
train_narre.py:124-125); `parallel: true` means one process per GPU with an RCCL gradient all-reduce
(launch with torch.distributed.run) instead of nn.DataParallel -- with nn.DataParallel's semantics: `batch_size` is the
GLOBAL batch, split evenly over the ranks (train_deepconn_pp.py:130 scatters each batch over the GPUs), every epoch
draws a new permutation (sampler.set_epoch), and validation shards the examples without padding so each one counts
once in the all-reduced RMSE; running loss / gnorm are accumulated on
the device and read back only at log lines (the reference calls loss.item() twice per step); the config key
`fast_step: true` switches to HipClipAdam and hipGraph replay of the step (train_step.py).
"""
from __future__ import annotations

import argparse
import datetime
import json
import math
import os
import time

import torch
import torch.nn as nn

from . import data as D
from . import functional as RF
from .train_step import GraphedForward, GraphedTrainStep, make_optimizer, train_step


class Args:
    """experiment.py:30-37 -- flat JSON -> attribute bag."""

    def __init__(self, cfg: dict):
        self.__dict__.update(cfg)


def parse_args(config_file: str) -> Args:
    with open(config_file) as f:
        return Args(json.load(f))


class EarlyStop(Exception):
    pass


DEFAULTS = dict(log_dir="logs", dataset="dataset", log=True, log_idx=500, verbose=False, parallel=False, epochs=64,
                batch_size=50, lr=0.002, max_grad_norm=5.0, patience=5, dropout=0.5, arch="CNN", use_pretrain=False,
                num_workers=0, fast_step=False, shuffle=True, seed=0, record_steps=False)


class _ShardSampler(torch.utils.data.Sampler):
    """Validation shard of one rank: examples rank, rank + world, ... in order -- no padding, so every example is seen by
    exactly one rank (DistributedSampler(drop_last=False) repeats examples to even the shards out, which would count them
    twice in the all-reduced squared error)."""

    def __init__(self, n: int, rank: int, world: int):
        self.idx = list(range(rank, n, world))

    def __iter__(self):
        return iter(self.idx)

    def __len__(self):
        return len(self.idx)


def make_loaders(train_set, valid_set, args, rank: int, world: int):
    """(train_loader, valid_loader, train_sampler, valid_sampler, per-rank batch) with nn.DataParallel's semantics on
    `world` ranks: `args.batch_size` is the global batch (train_deepconn_pp.py:130 scatters it over the GPUs), the training
    sampler reshuffles every epoch once the caller runs set_epoch(epoch), validation is sharded without padding."""
    parallel = world > 1
    if parallel and args.batch_size % world:
        raise ValueError(f"batch_size {args.batch_size} does not split evenly over {world} ranks")
    rank_batch = args.batch_size // world if parallel else args.batch_size
    train_sampler = valid_sampler = None
    if parallel:
        train_sampler = torch.utils.data.distributed.DistributedSampler(
            train_set, num_replicas=world, rank=rank, shuffle=bool(args.shuffle), seed=int(args.seed), drop_last=True)
        valid_sampler = _ShardSampler(len(valid_set), rank, world)
    gen = torch.Generator().manual_seed(int(args.seed))
    train_loader = torch.utils.data.DataLoader(
        train_set, batch_size=rank_batch, shuffle=bool(args.shuffle) and train_sampler is None, sampler=train_sampler,
        collate_fn=train_set.collate_fn, num_workers=args.num_workers, drop_last=train_sampler is not None, generator=gen)
    valid_loader = torch.utils.data.DataLoader(
        valid_set, batch_size=rank_batch, shuffle=False, sampler=valid_sampler, collate_fn=valid_set.collate_fn,
        num_workers=args.num_workers)
    return train_loader, valid_loader, train_sampler, valid_sampler, rank_batch


class ReviewExperiment:
    KINDS = ("deepconn", "narre", "dual_att", "simple_siamese")

    def __init__(self, kind: str, args: Args, reference_quirks: bool = False, uid: str | None = None):
        if kind not in self.KINDS:
            raise ValueError(f"{kind} is not one of {self.KINDS}")
        for k, v in DEFAULTS.items():
            if not hasattr(args, k):
                setattr(args, k, v)
        if getattr(args, "use_pretrain", False):
            raise RuntimeError("use_pretrain needs gensim + a word2vec file (train_deepconn_pp.py:105-119): pass "
                               "pretrained rows through the model's `pretrained_embeddings` argument instead")
        self.kind, self.args, self.quirks = kind, args, reference_quirks
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise RuntimeError("the HIP path needs an MI355X: there is no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        self.uid = uid or datetime.datetime.now().strftime("%y%m%d_%H%M%S")
        self.updates = 0
        self.best_rmse = 1e3
        self.patience = 0
        self.grad_sync = None

        review_split = kind in ("narre", "simple_siamese")
        cls = D.ReviewDataset if review_split else D.DocDataset
        kw = {} if review_split else {"with_ids": kind == "deepconn"}
        self.train_set = cls(args.data_dir, "train", **kw)
        self.valid_set = cls(args.data_dir, "valid", **kw)
        self._make_dir()
        self.build_model()
        # both splits were range-checked against their tables when they were loaded (data.validate_ranges): the per-forward
        # device-side check (one launch) is not needed on top
        self.model.validate_ids = False
        self.step_losses = [] if args.record_steps else None      # per-step training loss (device scalars), for tests
        self.parallel = self.world > 1 and bool(args.parallel)
        (self.train_loader, self.valid_loader, self.train_sampler, self.valid_sampler,
         self.rank_batch) = make_loaders(self.train_set, self.valid_set, args, self.rank, self.world if self.parallel else 1)
        # fast_step: clip + Adam as two HIP launches and the whole step replayed as a hipGraph for full-size batches
        # (same update rule; see train_step.HipClipAdam / GraphedTrainStep)
        self.optimizer = make_optimizer(self.model, lr=args.lr, hip_clip_adam=bool(args.fast_step))
        self._graphed = None
        self._graphed_key = None
        self.loss_func = nn.MSELoss()
        if self.parallel:
            from .distributed import GradAllReduce, broadcast_parameters, init_process_group_from_env
            init_process_group_from_env(os.environ.get("RBR_TRAINER_BACKEND", "nccl"))
            broadcast_parameters(self.model)
            self.grad_sync = GradAllReduce(self.model)
        self.print_args()
        self.print_model_stats()

    # ------------------------------------------------------------------ bookkeeping (experiment.py:64-123)
    def _make_dir(self):
        a = self.args
        self.out_dir = "./{}/{}/{}/{}".format(a.log_dir, a.dataset, a.model_name, self.uid)
        os.makedirs(self.out_dir, exist_ok=True)
        self.log_path = os.path.join(self.out_dir, "log.txt")

    def print_write_to_log(self, text: str):
        if self.rank != 0:
            return
        if self.args.log:
            try:
                with open(self.log_path, "a") as f:
                    f.write(text + "\n")
            except IOError:
                print("Cannot write a line into {}".format(self.log_path))
        print(text, flush=True)

    def print_args(self):
        for name, val in self.args.__dict__.items():
            self.print_write_to_log("{}: {}".format(name, val))
        self.print_write_to_log("=" * 50)

    def print_model_stats(self):
        self.print_write_to_log("List of all Trainable Variables")
        for i, (name, p) in enumerate(self.model.named_parameters()):
            if p.requires_grad:
                self.print_write_to_log("param {:3}: {:15} {}".format(i, str(tuple(p.shape)), name))
        n = sum(p.numel() for p in self.model.parameters())
        self.print_write_to_log("The total number of trainable parameters: {:,d}".format(n))
        self.print_write_to_log("=" * 50)

    def save(self, name: str):
        if self.rank != 0:
            return
        if not name.endswith(".pt"):
            name += ".pt"
        torch.save({"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "updates": self.updates,
                    "args": dict(self.args.__dict__)}, os.path.join(self.out_dir, name))

    # ------------------------------------------------------------------ model (train_*.py build_model)
    def build_model(self):
        a, ds = self.args, self.train_set
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            if self.kind == "deepconn":
                from .models.deepconn.deepconn import DeepCoNNpp
                ks = [3] if self.quirks else a.kernel_sizes
                self.model = DeepCoNNpp(user_size=ds.user_num, item_size=ds.item_num, vocab_size=ds.vocab_size, kernel_sizes=ks,
                                        hidden_dim=a.hidden_dim, embedding_dim=a.embedding_dim, dropout=a.dropout,
                                        latent_dim=a.latent_dim, doc_len=ds.doc_len, pretrained_embeddings=None, arch=a.arch)
            elif self.kind == "narre":
                from .models.narre.narre import NARRE
                ks = [3] if self.quirks else a.kernel_sizes
                hd = 150 if self.quirks else a.hidden_dim
                self.model = NARRE(user_size=ds.user_num, item_size=ds.item_num, vocab_size=ds.vocab_size, kernel_sizes=ks,
                                   hidden_dim=hd, embedding_dim=a.embedding_dim, att_dim=a.att_dim, latent_dim=a.latent_dim,
                                   max_doc_num=ds.rv_num, max_doc_len=ds.rv_len, dropout=a.dropout, word_padding_idx=0,
                                   user_padding_idx=0, item_padding_idx=0, pretrained_embeddings=None, arch=a.arch)
            elif self.kind == "simple_siamese":
                # trainer/train_simple_siamese.py:161-167 (keys of models/simple_siamese/defalut_simple_train.json)
                from .models.simple_siamese.simple_siamese import SimpleSiamese
                self.model = SimpleSiamese(embedding_dim=a.embedding_dim, latent_dim=a.latent_dim, vocab_size=ds.vocab_size,
                                           user_size=ds.user_num, item_size=ds.item_num, pretrained_embeddings=None,
                                           freeze_embeddings=getattr(a, "freeze_embeddings", False), dropout=a.dropout,
                                           word_dropout=getattr(a, "word_dropout", 0.2),
                                           review_dropout=getattr(a, "review_dropout", 0.0),
                                           use_ui_bias=getattr(a, "use_ui_bias", True),
                                           latent_transform=getattr(a, "latent_transform", False))
            else:
                from .models.dual_att.dual_att import DualAtt
                self.model = DualAtt(vocab_size=ds.vocab_size, doc_len=ds.doc_len, l_window_size=a.l_window_size,
                                     l_out_size=a.l_out_size, g_out_size=a.g_out_size, emb_size=a.emb_size,
                                     hidden_size_1=a.hidden_size_1, hidden_size_2=a.hidden_size_2, dropout=a.dropout,
                                     pretrained_embeddings=None)
        self.model.to(self.device)

    # ------------------------------------------------------------------ data
    def _to_device(self, batch):
        batch = [t.to(self.device, non_blocking=True) for t in batch]
        if self.kind == "simple_siamese":
            # review-split batch (u_revs, i_revs, word masks x2, ids x2, review ids x2, ratings) -> the model's arguments:
            # review masks mark the reviews that hold any token (simple_siamese/utils.py:94-106)
            u_revs, i_revs, u_wm, i_wm, u_ids, i_ids = batch[:6]
            return (u_revs, i_revs, u_wm, i_wm, u_wm.any(-1), i_wm.any(-1), u_ids, i_ids), batch[-1]
        return tuple(batch[:-1]), batch[-1]

    # ------------------------------------------------------------------ loops (train_deepconn_pp.py:143-232)
    def train_one_epoch(self, epoch: int):
        a = self.args
        loader = self.train_loader
        if self.train_sampler is not None:
            self.train_sampler.set_epoch(epoch)          # a new permutation every epoch, the same one on every rank
        loss_sum = torch.zeros((), device=self.device)
        sq_err = torch.zeros((), device=self.device)
        gnorm = torch.zeros((), device=self.device)
        steps = count = 0
        start = time.time()
        self.model.train()
        for i, batch in enumerate(loader):
            staged = self._step_from_host(batch)
            if staged is not None:
                loss, gnorm, ratings = staged
            else:
                inputs, ratings = self._to_device(batch)
                loss, gnorm = self._step(inputs, ratings)
            self.updates += 1
            if self.step_losses is not None:
                self.step_losses.append(loss.detach().clone())
            loss_sum += loss
            sq_err += loss * ratings.size(0)
            steps += 1
            count += ratings.size(0)
            if (i + 1) % a.log_idx == 0 and a.log:
                elapsed = (time.time() - start) / a.log_idx
                rmse = math.sqrt(float(sq_err) / count)       # the step's synchronisation point (the reference syncs every step)
                RF.check_id_errors(self.device)               # an id outside its table -> IndexError, as nn.Embedding raises
                self.print_write_to_log(
                    "epoch: {}/{}, step: {}/{}, loss: {:.3f}, rmse: {:.3f}, lr: {}, gnorm: {:3f}, time: {:.3f}".format(
                        epoch, a.epochs, i + 1, len(loader), float(loss_sum) / steps, rmse,
                        self.optimizer.param_groups[0]["lr"], float(gnorm), elapsed))
                loss_sum.zero_()
                sq_err.zero_()
                steps = count = 0
                start = time.time()

    def _step_from_host(self, batch):
        """`fast_step` with the recorded step in hand: a loader batch of the recorded shape goes host-to-device straight into
        the step's input block (GraphedTrainStep.stage) and the step is replayed -- no intermediate device tensors, no
        device-to-device copies.  None: not applicable (first batch, ragged batch, a model whose inputs are derived on the
        device), the caller takes the _to_device + _step route."""
        if not self.args.fast_step or self._graphed is None or self.kind == "simple_siamese":
            return None
        key = tuple((t.shape, t.dtype) for t in batch)
        if key != self._graphed_key:
            return None
        self._graphed.stage(0, tuple(batch[:-1]), batch[-1])
        loss, gnorm, _ = self._graphed(slot=0)
        return loss.clone(), gnorm.clone(), self._graphed.ratings

    def _step(self, inputs, ratings):
        """One optimisation step; with `fast_step` the step recorded for this batch shape is replayed (a ragged last
        batch, or a batch of another shape, runs eagerly)."""
        a = self.args
        if a.fast_step:
            key = tuple((t.shape, t.dtype) for t in inputs) + ((ratings.shape, ratings.dtype),)
            if self._graphed is None:
                self._graphed = GraphedTrainStep(self.model, self.optimizer, inputs, ratings, a.max_grad_norm, self.grad_sync)
                self._graphed_key = key
            if key == self._graphed_key:
                loss, gnorm, _ = self._graphed(inputs, ratings)
                return loss.clone(), gnorm.clone()          # the graph's outputs are overwritten by the next replay
        loss, gnorm, _ = train_step(self.model, self.optimizer, inputs, ratings, a.max_grad_norm, self.grad_sync)
        return loss, gnorm

    def _eval_forward(self, inputs):
        """The eval forward: replayed from a hipGraph for the loader's regular batch shape (recorded at first use; the
        parameters are read in place, so training steps in between need no re-recording), eager for any other shape
        (the ragged last batch) and when graphs are off."""
        if self.args.fast_step and inputs and inputs[0].is_cuda:
            g = getattr(self, "_graphed_eval", None)
            if g is None:
                try:
                    g = self._graphed_eval = GraphedForward(self.model, inputs)
                except Exception as e:          # a capture the runtime refuses costs the speed-up, not the validation
                    self.print_write_to_log(f"eval forward not graphed ({type(e).__name__}: {str(e)[:100]})")
                    g = self._graphed_eval = False
            if g and g.matches(inputs):
                return g(inputs)
        out = self.model(*inputs)
        return out[0] if isinstance(out, tuple) else out

    def valid_one_epoch(self):
        loader = self.valid_loader
        sq_err = torch.zeros((), device=self.device, dtype=torch.float64)
        loss_sum = torch.zeros((), device=self.device, dtype=torch.float64)
        count = torch.zeros((), device=self.device, dtype=torch.float64)
        steps = 0
        self.model.eval()
        with torch.no_grad():
            for batch in loader:
                inputs, ratings = self._to_device(batch)
                pred = self._eval_forward(inputs)
                loss = self.loss_func(pred, ratings)
                sq_err += loss.double() * ratings.size(0)
                loss_sum += loss.double()
                count += ratings.size(0)
                steps += 1
        if self.parallel:
            import torch.distributed as dist
            for t in (sq_err, count):
                dist.all_reduce(t)
        self.valid_count = int(count)                     # examples that entered the RMSE: len(valid_set), each once
        rmse = math.sqrt(float(sq_err) / max(float(count), 1.0))
        self.last_valid_rmse = rmse
        RF.check_id_errors(self.device)
        if rmse < self.best_rmse:
            self.best_rmse = rmse
            self.save("best_model.pt")
            self.patience = 0
        else:
            self.patience += 1
        self.print_write_to_log("valid loss: {:.3f}, valid rmse: {:.3f}, best rmse: {:.3f}".format(
            float(loss_sum) / max(steps, 1), rmse, self.best_rmse))
        if self.patience >= self.args.patience:
            raise EarlyStop("early stop")

    def train(self):
        self.print_write_to_log("start training ...")
        for epoch in range(self.args.epochs):
            self.train_one_epoch(epoch)
            self.valid_one_epoch()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", required=True, choices=ReviewExperiment.KINDS)
    ap.add_argument("--config", required=True, help="flat JSON config with the reference's keys")
    ap.add_argument("--reference-quirks", action="store_true", help="hard-code kernel_sizes=[3] (and NARRE hidden_dim=150)")
    a = ap.parse_args(argv)
    exp = ReviewExperiment(a.model, parse_args(a.config), reference_quirks=a.reference_quirks)
    try:
        exp.train()
    except EarlyStop:
        exp.print_write_to_log("early stop (patience {})".format(exp.args.patience))


if __name__ == "__main__":
    main()
