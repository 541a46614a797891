"""YAML-driven training harness around the hot path: the reference's experiment plumbing
(utils/train_util.py:19-134) and the skeleton of its training scripts (train_segmentation.py:64-230,
train_classification.py) with synthetic loaders, so that a reference config + model file runs end to end on this
package's layers and checkpoints stay interchangeable.

* config: the reference's YAML schema — `experiment.{root,writer_root}`, `data.{batch_size,num_points,...}`,
  `model.generator` (a python file defining `Model`; the remaining `model.*` keys are its kwargs),
  `train.{optimizer,scheduler,num_epochs,save_each,...}`, optional `restore.{generator,optimizer,new_lr}`.
* checkpoints: `<exp>/<name>_<epoch_name>_<n>.t7` holding a plain `state_dict()` (train_util.py:74-80); parameter names of
  this package's modules equal the reference's, so released weights load with `strict=True`; `restore_exp_fix` drops the
  `module.` prefix DistributedDataParallel adds (train_util.py:98-117).
* data: `SyntheticClouds` produces batches of the shapes and dtypes the reference loaders yield (no files needed); with
  `data.kind` in the config (or `Trainer(dataset=...)`) the loop runs on `data/datasets.py`'s readers instead — `scanobjectnn`
  (datasets/scanobjectnn.py: `data.path` = the .h5 / .npz file) and `s3dis` (datasets/s3dis_v2.py: `data.path` = the
  indoor3d_sem_seg_hdf5_data directory), items equal to the reference loaders' under the same seeds.  The remaining dataset
  variants (s3dis_closer*, GRNet completion, image_point) stay out of scope.
"""
import copy
import datetime
import shutil
import time
from collections import OrderedDict
from pathlib import Path

import torch
from torch import nn

from . import parallel


def worker_init_fn(worker_id):
    import random
    import numpy as np
    seed = int(np.random.get_state()[1][0]) + worker_id
    np.random.seed(seed % (2 ** 32))
    random.seed(seed)


def get_model(model_file, params_dict, exp_dir=None):
    """Instantiate `Model(**params_dict)` from a model python file; keep a copy beside the experiment."""
    env = {"__name__": "model_file"}
    with open(str(model_file), "r") as f:
        exec(compile(f.read(), str(model_file), "exec"), env)
    model = env["Model"](**params_dict)
    if exp_dir is not None:
        assert Path(exp_dir).exists()
        shutil.copy2(str(model_file), str(exp_dir))
    return model


def check_model_paths(*paths):
    out = []
    for p in paths:
        f = Path(p)
        assert f.exists() and f.suffix == ".py", p
        out.append((p, f.name[:-3]))
    return out


class NullWriter:
    """Stands in for tensorboardX.SummaryWriter when it is not installed: keeps the last value of every tag."""

    def __init__(self, logdir=None):
        self.logdir, self.scalars = logdir, {}

    def add_scalar(self, tag, value, global_step=None):
        self.scalars[tag] = (float(value), global_step)

    def close(self):
        pass


def _summary_writer(path):
    try:
        from tensorboardX import SummaryWriter
    except ImportError:
        try:
            from torch.utils.tensorboard import SummaryWriter
        except ImportError:
            return NullWriter(str(path))
    return SummaryWriter(str(path))


def create_experiment(*desc, params_dict):
    exp_path, writer_path = Path(params_dict["exp_root"]), Path(params_dict["writer_root"])
    assert writer_path.exists(), writer_path
    full_desc = "_".join([*desc, datetime.datetime.now().strftime("%d_%m_%y_%H_%M_%S")])
    writer = _summary_writer(writer_path.joinpath(full_desc))
    exp_dir = exp_path.joinpath(full_desc)
    exp_dir.mkdir(parents=True)
    if "config_path" in params_dict:
        cfg = Path(params_dict["config_path"])
        assert cfg.exists()
        shutil.copy2(str(cfg), str(exp_dir))
    return writer, exp_dir, full_desc


def save_exp(objects, names, exp_path, epoch, epoch_name="epoch"):
    assert len(objects) == len(names)
    for obj, name in zip(objects, names):
        with open("{}/{}_{}_{}.t7".format(str(exp_path), name, epoch_name, epoch), "wb") as f:
            torch.save(obj.state_dict(), f)


def restore_exp(objects, names, device, verbose=True, strict=True):
    assert len(objects) == len(names)
    for obj, name in zip(objects, names):
        assert Path(name).exists(), name
        if verbose:
            print("restoring from {}".format(name))
        with open(name, "rb") as f:
            state = torch.load(f, map_location=device)
        if isinstance(obj, nn.Module):
            obj.load_state_dict(state, strict=strict)
        else:
            obj.load_state_dict(state)


def restore_exp_fix(objects, names, device=None, verbose=True):
    """Load checkpoints written from a DistributedDataParallel wrapper into plain modules (strict)."""
    device = device if device is not None else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    assert len(objects) == len(names)
    for obj, name in zip(objects, names):
        assert Path(name).exists(), name
        if verbose:
            print("restoring from {}".format(name))
        with open(name, "rb") as f:
            state = torch.load(f, map_location=device)
        obj.load_state_dict(OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state.items()), strict=True)


def make_optimizer(params, opt_cfg):
    cfg = dict(opt_cfg)                        # (the reference deletes `type` from the caller's dict; a copy keeps cfg reusable)
    return getattr(torch.optim, cfg.pop("type"))(params, **cfg)


def make_scheduler(optimizer, scheduler_cfg):
    cfg = dict(scheduler_cfg)
    return getattr(torch.optim.lr_scheduler, cfg.pop("type"))(optimizer, **cfg)


class SyntheticClouds(torch.utils.data.Dataset):
    """Batches shaped like the reference loaders': `segmentation` -> (points f32 [N, 3], labels i64 [N]) as
    datasets/s3dis_v2.py yields; `classification` -> (points f32 [N, 3], label i64 [], mask f32 [N]) as
    datasets/scanobjectnn.py yields; `completion` -> (noise f32 [4, N]: points on the unit sphere + the "is a real point"
    label, partial cloud f32 [n_classes, 3] (n_classes = its size), ground truth f32 [N, 3]) — what train_inpainter.py:178-185
    hands to the generator and the losses after `partial_postproces`.  Deterministic per index."""

    def __init__(self, task, num_points, n_classes, length=64, seed=0, channels=3):
        assert task in ("segmentation", "classification", "completion") and channels >= 3
        self.task, self.n, self.k, self.length, self.seed, self.ch = task, num_points, n_classes, length, seed, channels

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        if self.task == "completion":
            gt = torch.nn.functional.normalize(torch.randn(self.n, 3, generator=g), dim=1) * (0.5 + 0.4 * torch.rand((), generator=g))
            part = gt[torch.randperm(self.n, generator=g)[: self.k]] + 0.01 * torch.randn(self.k, 3, generator=g)
            sphere = torch.nn.functional.normalize(torch.randn(3, self.n, generator=g), dim=0)
            noise = torch.cat([sphere, (torch.rand(1, self.n, generator=g) > 0.5).float()], dim=0)
            return noise, part, gt
        pts = torch.rand(self.n, self.ch, generator=g) * 2 - 1          # xyz (+ colour / normalised position channels)
        if self.task == "segmentation":
            # labels follow the geometry (octants), so a few steps of training move the loss
            lab = ((pts[:, 0] > 0).long() + 2 * (pts[:, 1] > 0).long() + 4 * (pts[:, 2] > 0).long()) % self.k
            return pts, lab
        lab = torch.randint(self.k, (), generator=g)
        pts = pts * (0.5 + 0.5 * (lab.float() + 1) / self.k)
        return pts, lab, (pts[:, 2] > 0).float()


def make_dataset(cfg, task, n_classes, length=64, channels=3, train=True):
    """The dataset `data.kind` of the config names: "synthetic" (default), "scanobjectnn" or "s3dis" (data/datasets.py)."""
    data = cfg["data"]
    kind = str(data.get("kind", "synthetic")).lower()
    if kind == "synthetic":
        return SyntheticClouds(task, data["num_points"], n_classes, length=length, channels=channels)
    from .data import datasets as D
    if kind == "scanobjectnn":
        assert task == "classification", "ScanObjectNN items are (points, label, mask): the classification task"
        return D.ScanObjectNN(data["path"], train=train, subsample=data.get("num_points"), center=data.get("center", True),
                              normalize=data.get("normalize", True))
    if kind == "s3dis":
        assert task == "segmentation", "S3DIS blocks are (points, labels): the segmentation task"
        return D.Indoor3DSemSeg(data["path"], data["num_points"], train=train, aug=bool(data.get("aug", train)),
                                test_area=data.get("test_area", "Area_5"), data_precent=float(data.get("data_precent", 1.0)))
    raise ValueError("data.kind must be synthetic, scanobjectnn or s3dis (got %r)" % kind)


class Trainer:
    """The reference scripts' loop: model file + YAML config -> DDP(+SyncBN) model, optimizer, scheduler, steps with
    loss reduction to rank 0, `.t7` checkpoints every `train.save_each` iterations.

    `task`: "segmentation" (loss = CE(pred[:, :, 0], labels), train_segmentation.py:178), "classification"
    (loss = CE(logits, label) + seg_weight * BCE-with-logits(mask), train_classification.py) or "completion"
    (loss = mean sqrt(EMD(rec, gt, 0.005, 50)) + chamfer_weight * loss_chamfer(rec, gt), train_inpainter.py:186-192;
    `n_classes` is then the size of the partial cloud)."""

    def __init__(self, cfg, task, n_classes, device=None, dist=None, exp_name="exp", dataset_length=64, make_dirs=True,
                 channels=3, dataset=None):
        self.cfg = cfg = copy.deepcopy(cfg)
        self.task, self.dist = task, dist
        self.rank = dist.get_rank() if parallel._active(dist) else 0
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        model_cfg = dict(cfg["model"])
        model_file, _ = check_model_paths(model_cfg.pop("generator"))[0]
        self.writer, self.exp_dir = NullWriter(), None
        if self.rank == 0 and make_dirs:
            Path(cfg["experiment"]["writer_root"]).mkdir(parents=True, exist_ok=True)
            params = {"exp_root": cfg["experiment"]["root"], "writer_root": cfg["experiment"]["writer_root"]}
            if "config_path" in cfg:
                params["config_path"] = cfg["config_path"]
            self.writer, self.exp_dir, _ = create_experiment(exp_name, params_dict=params)
        model = get_model(model_file, model_cfg, exp_dir=self.exp_dir).to(self.device)
        if self.device.type == "cuda":
            from .layers.pointwise import convert_pointwise
            convert_pointwise(model)      # the model file's plain nn.Conv1d(k=1) stems / heads onto the blocks' GEMM kernels
        if "restore" in cfg and "generator" in cfg["restore"]:
            restore_exp_fix([model], [cfg["restore"]["generator"]], device=self.device, verbose=self.rank == 0)
        self._stream = None
        if parallel._active(dist):
            if self.device.type == "cuda":
                # DDP's gradient hooks (and its bucketed all-reduce) run on the stream the wrapper was BUILT on: the training
                # steps, and the capture of fit(hip_graph=True), use that same side stream (torch notes/cuda.rst, "Usage
                # with DistributedDataParallel")
                self._stream = torch.cuda.Stream(device=self.device)
                self._stream.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._stream):
                    model = parallel.data_parallel(model, self.device.index)
            else:
                model = parallel.data_parallel(model, None)
        self.model = model
        tr = cfg["train"]
        if "scale_lr" in tr:      # the learnable residual scales of the AdaIN blocks get their own rate
            named = list(model.named_parameters())
            params = [{"params": [p for n, p in named if not n.endswith("scale")]},
                      {"params": [p for n, p in named if n.endswith("scale")], "lr": tr["scale_lr"]}]
        else:
            params = model.parameters()
        self.optimizer = make_optimizer(params, tr["optimizer"])
        if "restore" in cfg and "optimizer" in cfg["restore"]:
            restore_exp([self.optimizer], [cfg["restore"]["optimizer"]], self.device, verbose=self.rank == 0)
            if "new_lr" in cfg["restore"]:
                for g in self.optimizer.param_groups:
                    g["lr"] = cfg["restore"]["new_lr"]
        self.scheduler = make_scheduler(self.optimizer, tr["scheduler"]) if "scheduler" in tr else None
        # `dataset`: any torch Dataset whose items have the task's layout; else what `data.kind` of the config names
        data = dataset if dataset is not None else make_dataset(cfg, task, n_classes, length=dataset_length, channels=channels)
        self.sampler = torch.utils.data.distributed.DistributedSampler(data) if parallel._active(dist) else None
        self.loader = torch.utils.data.DataLoader(data, batch_size=cfg["data"]["batch_size"], shuffle=self.sampler is None,
                                                  num_workers=int(cfg["data"].get("num_workers", 0)), sampler=self.sampler,
                                                  drop_last=bool(cfg["data"].get("drop_last", True)),
                                                  worker_init_fn=worker_init_fn)
        self.ce, self.bce = nn.CrossEntropyLoss(), nn.BCEWithLogitsLoss()
        self.iters = 0

    def _loss(self, batch):
        if self.task == "segmentation":
            pts, labels = batch
            pcd = pts.permute(0, 2, 1)[:, :, None].to(self.device)                    # (B, channels, 1, N)
            out = self.model(pcd)
            pred = out[0] if isinstance(out, (tuple, list)) else out                  # the reference returns (pred, lattice stats)
            return self.ce(pred[:, :, 0], labels.to(self.device))
        if self.task == "completion":
            from .chamfer import loss_chamfer
            from .emd import emdModule
            noise, part, gt = batch
            out = self.model(noise.to(self.device), part.permute(0, 2, 1)[:, :, None].to(self.device))
            rec = out[0] if isinstance(out, (tuple, list)) else out                   # (B, 3, 1, N)
            gt4 = gt.permute(0, 2, 1)[:, :, None].to(self.device)
            tr = self.cfg["train"]
            dist, _ = emdModule()(rec[:, :, 0].permute(0, 2, 1), gt4[:, :, 0].permute(0, 2, 1),
                                  float(tr.get("emd_eps", 0.005)), int(tr.get("emd_iters", 50)))
            return torch.sqrt(dist).mean(1).mean() + float(tr.get("chamfer_weight", 1.0)) * loss_chamfer(rec, gt4)
        pts, label, mask = batch
        logits, mask_pred = self.model(pts.permute(0, 2, 1)[:, :, None].to(self.device))
        w = float(self.cfg["train"].get("seg_weight", 0.5))
        return (self.ce(logits, label.to(self.device).long())
                + w * self.bce(mask_pred.reshape(mask.shape[0], -1), mask.to(self.device).float()))     # (ScanObjectNN's mask is int64)

    def save(self):
        if self.rank == 0 and self.exp_dir is not None:
            parallel.save_exp_parallel([self.model, self.optimizer], ["generator", "g_opt"], exp_path=self.exp_dir,
                                       epoch=self.iters, epoch_name="iter")

    # -- one training step -------------------------------------------------------------------------------------------
    def _eager_step(self, batch):
        loss = self._loss(batch)
        loss.backward()
        self.optimizer.step()
        self.optimizer.zero_grad()
        return loss.detach()

    # DistributedDataParallel needs this many eager iterations before a capture (its reducer rebuilds the buckets after the
    # first one and settles its bookkeeping; torch notes/cuda.rst "Usage with DistributedDataParallel")
    DDP_WARMUP = 11

    def _capture(self, batch):
        """forward + loss + backward of one batch shape as ONE HIP graph on static input buffers (every libcloudct launch
        goes to torch's current stream, so the whole step captures — under DistributedDataParallel with its bucketed
        gradient all-reduce and the norms' statistics exchanges: RCCL collectives are captured like kernels); the optimizer
        step stays outside.  The blocks' eager launch rate is host-bound (~250 launches per block through Python / ctypes:
        the segmenter step 30.8 ms eager vs 24.8 ms graphed, profiles/r3_tools_output.txt).  Returns the record of the
        capture: (graph, static inputs, static loss, the gradient tensors the graph writes)."""
        static = [t.to(self.device).clone() for t in batch]
        buffers = [b.clone() for b in self.model.buffers()]          # the warm-up passes must not count as training steps
        params = [p for p in self.model.parameters() if p.requires_grad]
        side = self._stream if self._stream is not None else torch.cuda.Stream()      # (DDP: the stream it was built on)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.DDP_WARMUP if parallel._active(self.dist) else 2):
                self.optimizer.zero_grad(set_to_none=True)
                self._loss(static).backward()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for b, saved in zip(self.model.buffers(), buffers):
                b.copy_(saved)
        self.optimizer.zero_grad(set_to_none=True)                   # the graph creates its own gradient tensors
        if parallel._active(self.dist):
            parallel.barrier(self.dist)
            parallel.quiesce()                                       # the process group's watchdog drops the warm-up's work items
        graph = torch.cuda.CUDAGraph()
        # (captured on the stream the warm-up ran on — under DDP the wrapper's, else `side`: the per-stream workspaces and
        #  tickets of the raster ops were created by the warm-up, outside the capture and outside the graph's private pool)
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            static_loss = self._loss(static)
            static_loss.backward()
        return graph, static, static_loss, [p.grad for p in params]

    def _graph_step(self, batch):
        """One step by graph replay.  A graph is captured per batch SHAPE (a ragged last batch, `data.drop_last: false`,
        gets its own); every graph owns the gradient tensors it writes, so the parameters' `.grad` are pointed at the
        replayed graph's before the optimizer steps — stepping with another graph's (stale) gradients, or dropping the
        captured tensors with zero_grad(), would silently stop the training."""
        key = tuple(tuple(t.shape) for t in batch)
        rec = self._graphs.get(key)
        if rec is None:
            rec, failure = None, None
            try:
                rec = self._capture(batch)
            except RuntimeError as ex:
                # only what a refused CAPTURE raises (HIP's stream-capture errors, e.g. the process group's watchdog touching an
                # event of the capturing stream); a genuine error inside the model or the loss is not swallowed as "capture failed"
                if not _is_capture_error(ex):
                    raise
                failure = ex
                torch.cuda.synchronize(self.device)
            # every rank takes the SAME path: one replaying a graph beside others stepping eagerly would deadlock in the reducer
            if parallel._active(self.dist):
                ok = torch.tensor([0 if failure is not None else 1], device=self.device, dtype=torch.int32)
                self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and failure is None:
                    failure, rec = RuntimeError("HIP-graph capture failed on another rank"), None
                if failure is not None:
                    # a capture that failed inside backward leaves DistributedDataParallel's reducer mid-iteration ("Expected to have
                    # finished reduction ..." on the next forward): there is no clean eager state to fall back to under DDP
                    raise RuntimeError("harness: HIP-graph capture failed under DistributedDataParallel (%s); rerun with "
                                       "train.hip_graph: false" % (failure,)) from failure
            if failure is not None:
                if self.rank == 0:
                    print("harness: HIP-graph capture failed (%s); continuing with eager steps" % (failure,), flush=True)
                rec = False
            self._graphs[key] = rec
        if rec is False:
            return self._eager_step(batch)
        graph, static, static_loss, grads = rec
        for dst, src in zip(static, batch):
            dst.copy_(src, non_blocking=True)
        graph.replay()
        for p, g in zip((p for p in self.model.parameters() if p.requires_grad), grads):
            p.grad = g
        self.optimizer.step()
        return static_loss.detach().clone()

    def fit(self, max_iters=None, hip_graph=None, log_each=None):
        """Runs `train.num_epochs` epochs (or `max_iters` steps); returns the rank-0 loss history.

        `hip_graph` (default: `train.hip_graph` of the config, else False): replay forward + loss + backward as one HIP graph
        (under DistributedDataParallel too: the gradient all-reduce and the norms' statistics exchanges are captured with
        the kernels, after DDP's warm-up iterations; the process group must have been created with
        TORCH_NCCL_ASYNC_ERROR_HANDLING=0 in the environment — launch.rank_env(capture=True) / spawn_ranks(capture=True) set it,
        torchrun users set it themselves; the price is that a hung collective then hangs the job instead of aborting it.  A capture that
        fails falls back to eager steps on one device and RAISES under DistributedDataParallel — the ranks agree on it first).  Losses stay on the
        device and are read back every `log_each` steps (default `train.log_each`, else 10) in ONE transfer — no host
        synchronisation per step (the reference reads the loss every step: train_segmentation.py:180-186)."""
        tr = self.cfg["train"]
        use_graph = bool(tr.get("hip_graph", False) if hip_graph is None else hip_graph)
        log_each = int(tr.get("log_each", 10) if log_each is None else log_each)
        self._graphs = {}
        history, pending = [], []

        def flush():
            if self.device.type == "cuda":
                from . import ops
                ops.mhct_core_check()        # (the one place the loop waits for the device anyway: a cluster timeout raises here)
            if pending and self.rank == 0:
                stamps, vals = zip(*pending)
                for it, v in zip(stamps, torch.stack(vals).tolist()):      # one device-to-host copy for the interval
                    history.append(v)
                    self.writer.add_scalar("train/loss", v, global_step=it)
            pending.clear()

        for epoch in range(tr["num_epochs"]):
            if self.sampler is not None:
                self.sampler.set_epoch(epoch)
            self.model.train()
            end = time.time()
            for batch in self.loader:
                if self._stream is not None:            # under DDP every step runs on the wrapper's stream (see __init__)
                    self._stream.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(self._stream):
                        loss = self._graph_step(batch) if use_graph else self._eager_step(batch)
                    torch.cuda.current_stream(self.device).wait_stream(self._stream)
                else:
                    loss = self._graph_step(batch) if use_graph else self._eager_step(batch)
                if self.scheduler is not None:
                    self.scheduler.step()
                reduced = parallel.reduce_loss_dict(self.dist, {"loss": loss})
                pending.append((self.iters, reduced["loss"].detach().reshape(())))
                self.iters += 1
                if self.iters % log_each == 0:
                    flush()
                    if self.rank == 0:
                        self.writer.add_scalar("train/batch_time", (time.time() - end) / log_each, global_step=self.iters)
                    end = time.time()
                if self.iters % tr.get("save_each", 1 << 62) == 0:
                    self.save()
                if max_iters is not None and self.iters >= max_iters:
                    flush()
                    return history
        flush()
        return history


def _is_capture_error(ex):
    """Is this RuntimeError HIP refusing or invalidating a stream capture (as opposed to an error of the captured code)?"""
    msg = str(ex).lower()
    return any(k in msg for k in ("capture", "hipgraph", "cudagraph", "graph", "operation not permitted when stream is capturing",
                                  "streamcapture"))


def load_config(path):
    import yaml
    with open(path, "r") as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    cfg["config_path"] = str(path)
    return cfg
