"""Mirror of the reference's ``train_encodersKIT.py`` for the path in scope: ``trainer`` (:45-249),
``selectProxiesByTriagulation`` (:252-284), ``collate_fn_PK`` (:289-290), ``samplePKBatches`` (:292-403, PK sampling
and clean/turbulence pairing; decode + augmentation go through the pluggable loader of daliid_amd.getFeatures).

One epoch of ``trainer.train`` = full-train-set inference -> class centers + 5 farthest-point proxies per class ->
one pass of PK batches {forward, L2-normalise(+1e-9), center + lambda*proxy loss, backward, Adam, EMA, sum ||theta||^2}.
The hot loop issues only HIP kernels (C ABI) and, with a process group, two RCCL collectives per step; nothing
synchronises with the host until the epoch summary is printed."""
import numpy as np
import torch

from . import _lib, ops_eval, optim, parallel
from .getFeatures import extractFeatures, get_image_loader
from .losses import LossHeads, _codes, _sample_weights

np.random.seed(12)
torch.manual_seed(12)

_train_loader = None


def set_train_loader(fn):
    """fn(paths, img_height, img_width, turb=None) -> float tensor [n,3,H,W]: the loader ``samplePKBatches`` uses, i.e.
    decode + the training transform of train_encodersKIT.py:313-320 (``daliid_amd.transforms.gpu_train_loader`` runs it
    on the GPU and is the default).  A loader that does NOT augment must be installed explicitly: falling back to the
    evaluation transform silently would train a different model than the reference."""
    global _train_loader
    _train_loader = fn


def get_train_loader():
    """The installed loader, else ``transforms.gpu_train_loader``: host decode, then Resize / RandomCrop(pad 10) /
    RandomHorizontalFlip / ColorJitter / RandomErasing / Normalize (train_encodersKIT.py:313-320) on the GPU."""
    if _train_loader is not None:
        return _train_loader
    from .transforms import gpu_train_loader
    return gpu_train_loader


def selectProxiesByTriagulation(X, num_proxies=5):
    """train_encodersKIT.py:252-284: farthest-point sampling; first pick ``np.random.choice(n)``, then repeatedly the
    point whose minimum distance to the chosen set is largest (``argsort(...)[-1]``).
    -> (indices LongTensor, max pairwise distance among the chosen).  One launch of dali_class_targets."""
    n = X.shape[0]
    dev = X.device
    first = torch.tensor([int(np.random.choice(n))], dtype=torch.int32, device=dev)
    order = torch.arange(n, dtype=torch.int32, device=dev)
    bounds = torch.tensor([0, n], dtype=torch.int32, device=dev)
    _, _, rows, max_dist = ops_eval.class_targets(X.contiguous().float(), order, bounds, first, num_proxies)
    rows = rows[rows >= 0].long()
    return rows, max_dist.item()


def build_centers_and_proxies(fvs, labels, num_proxies=5, process_group=None):
    """train_encodersKIT.py:113-156 on device-resident features: per class, proxies by farthest-point sampling and
    the center = mean of the un-normalised embeddings; both L2-normalised (no epsilon).  The per-class Python loop of
    the reference becomes one kernel launch (one block per identity); the host only sorts the labels and draws the
    first pick of every identity from numpy's global stream, one ``np.random.choice(n)`` per identity in label order as
    the reference does (:257).
    -> (centers, centers_labels, all_proxies, proxies_labels, mean_max_distance)"""
    labels = np.asarray(labels)
    centers_labels, counts = np.unique(labels, return_counts=True)
    order = np.argsort(labels, kind="stable").astype(np.int32)
    bounds = np.concatenate(([0], np.cumsum(counts))).astype(np.int32)
    first = np.array([np.random.choice(int(n)) for n in counts], dtype=np.int32)
    if process_group is not None:          # one process draws in the reference; the ranks' numpy streams have diverged in the PK loop
        first = parallel.broadcast_from_rank0(first, process_group)
    dev = fvs.device
    centers, proxies, rows, max_dist = ops_eval.class_targets(
        fvs.contiguous().float(), torch.from_numpy(order).to(dev), torch.from_numpy(bounds).to(dev), torch.from_numpy(first).to(dev),
        num_proxies)
    per_class = np.minimum(counts, num_proxies)
    if (per_class < num_proxies).any():                      # identities with fewer images than proxies: drop the empty slots
        keep = torch.from_numpy((np.arange(num_proxies)[None, :] < per_class[:, None]).reshape(-1)).to(dev)
        proxies = proxies[keep].contiguous()
    proxies_labels = np.repeat(centers_labels, per_class)
    return centers, centers_labels, proxies, proxies_labels, max_dist.mean().item()


def collate_fn_PK(batch):
    return batch


class samplePKBatches:
    """train_encodersKIT.py:292-403.  ``__getitem__(idx)`` -> (images [k or 2k,3,H,W], labels, distortion levels) for
    the idx-th identity of a shuffled identity list: up to K images of that identity; with ``kind_of_transform == 1``
    every image is paired with a turbulence-distorted copy of random strength 1..5 (distortion level = strength)."""

    def __init__(self, dataset, images, labels, img_height, img_width, turbulance_dir_path, kind_of_transform, K=4, turb_strength=0):
        self.images_names = images[:, 0]
        # column 3 = kind of record (datasetUtils.py:15); only 'person' rows are ever put in a batch (train_encodersKIT.py:299, :348).
        # Record arrays without that column (in-memory synthetic sets) are all persons.
        self.reid_instances = images[:, 3] if images.shape[1] > 3 else None
        self.labels = np.asarray(labels)
        self.labels_set = np.unique(labels)
        self.K = K
        np.random.shuffle(self.labels_set)
        self.img_height, self.img_width = img_height, img_width
        self.dataset, self.turbulance_dir_path, self.kind_of_transform = dataset, turbulance_dir_path, kind_of_transform

    def __len__(self):
        return len(self.labels_set)

    def _select(self, pid):
        """The identity's files and the up-to-K picks among them (train_encodersKIT.py:329-339): the draw runs over ALL rows of the identity,
        then rows whose kind is not 'person' are dropped without drawing anything for them (:348: the whole body sits under that test)."""
        rows = self.labels == pid
        names = self.images_names[rows]
        sel = np.random.choice(names.shape[0], size=min(names.shape[0], self.K), replace=False)
        if self.reid_instances is not None:
            sel = sel[self.reid_instances[rows][sel] == "person"]
            if len(sel) == 0:              # the reference dies here too (``[].shape``, :397): say why
                raise _lib.DaliError("samplePKBatches: identity %r has no 'person' record among its picks" % (pid,))
        return names, sel

    def __getitem__(self, idx):
        pid = self.labels_set[idx]
        names, sel = self._select(pid)
        loader = get_train_loader()
        clean = loader(list(names[sel]), self.img_height, self.img_width, None)
        if self.kind_of_transform == 0:
            imgs, dist = clean, np.zeros(len(sel), dtype=np.int32)
        else:
            strengths = np.random.choice([1, 2, 3, 4, 5], size=len(sel))
            turb = torch.cat([loader([names[s]], self.img_height, self.img_width, (self.turbulance_dir_path, int(t), self.dataset))
                              for s, t in zip(sel, strengths)], 0)
            imgs = torch.stack((clean, turb.to(clean.device)), 1).flatten(0, 1)          # clean, distorted, clean, distorted ...
            dist = np.stack((np.zeros(len(sel), dtype=np.int32), strengths.astype(np.int32)), 1).reshape(-1)
        return imgs, torch.ones(imgs.shape[0]) * float(pid), dist

    def plan(self, idx, loader):
        """``__getitem__`` with the image work deferred: the same numpy / torch draws in the same order (selection, the clean images'
        augmentation parameters, strengths, each distorted image's parameters), but the files are only LISTED -- in the final
        (clean, distorted, clean, distorted ...) order -- so that a whole PK batch is decoded on the pool and resized + augmented by one
        launch each (``plan_batch`` / ``finish_batch``).  -> (ImagePlan, labels tensor, distortion levels)"""
        pid = self.labels_set[idx]
        names, sel = self._select(pid)
        clean = loader.plan(list(names[sel]), self.img_height, self.img_width, None)
        if self.kind_of_transform == 0:
            plan, dist = clean, np.zeros(len(sel), dtype=np.int32)
        else:
            strengths = np.random.choice([1, 2, 3, 4, 5], size=len(sel))
            turb = [loader.plan([names[s]], self.img_height, self.img_width, (self.turbulance_dir_path, int(t), self.dataset))
                    for s, t in zip(sel, strengths)]
            k = len(sel)
            order = [j for i in range(k) for j in (i, k + i)]                          # clean_0, turb_0, clean_1, turb_1, ...
            plan = clean.concat([clean] + turb, order)                                 # (the plan type's own concat: transforms.ImagePlan.concat)
            dist = np.stack((np.zeros(k, dtype=np.int32), strengths.astype(np.int32)), 1).reshape(-1)
        return plan, torch.ones(len(plan.files)) * float(pid), dist

    def plan_batch(self, ids, loader):
        """Plans of the identities ``ids`` merged into one and submitted to the decode pool.  -> ticket for ``finish_batch``"""
        parts = [self.plan(i, loader) for i in ids]
        ticket = loader.submit(parts[0][0].concat([p[0] for p in parts]))
        return ticket, torch.cat([p[1] for p in parts], 0), np.concatenate([p[2] for p in parts])

    @staticmethod
    def finish_batch(planned, loader, dev):
        ticket, labels, dist = planned
        return loader.finish(ticket, dev), labels, dist


class trainer(object):
    """train_encodersKIT.py:45-249, same constructor arguments and attributes."""

    def __init__(self, dataset, selected_images, model_name, labels_dict, img_height, img_width, turbulance_dir_path, is_clean_training,
                 kind_of_transform, optimizer, P, K, tau, beta, lambda_proxy, number_of_epoches, model_online, model_momentum, gpu_indexes,
                 version, process_group=None):
        self.dataset, self.selected_images, self.model_name, self.labels_dict = dataset, selected_images, model_name, labels_dict
        self.img_height, self.img_width = img_height, img_width
        self.turbulance_dir_path, self.is_clean_training, self.kind_of_transform = turbulance_dir_path, is_clean_training, kind_of_transform
        self.num_proxies = 5
        self.optimizer = optimizer
        self.P, self.K, self.tau, self.beta, self.lambda_proxy = P, K, tau, beta, lambda_proxy
        self.number_of_epoches = number_of_epoches
        self.model_online, self.model_momentum = model_online, model_momentum
        self.gpu_indexes, self.version = gpu_indexes, version
        self.process_group = process_group
        self._net = getattr(model_online, "module", model_online)
        self._mom = getattr(model_momentum, "module", model_momentum)
        self._adam = optimizer if isinstance(optimizer, optim.FusedAdam) else optim.FusedAdam.from_torch(optimizer, self._net)
        self._dp = None                      # parallel.GradReducer, built at the first step (needs the plan's stage ranges)
        self.prefetch_depth = 2              # PK batches planned + decoding ahead of the one being trained on (batched loaders only)
        self.last_epoch_stats = None

    # ---- epoch-level: inference over the train set, centers, proxies (train_encodersKIT.py:104-156) ----
    def extract_train_features(self, selected_images):
        """train_encodersKIT.py:104-110.  With a process group the train set is cut into one contiguous slice per rank (nn.DataParallel
        splits every 500-image batch over the GPUs instead, getFeatures.py:56-67; eval-mode features do not depend on the split) and one
        all-gather gives every rank all rows, in dataset order: all ranks then build the same centers and proxies."""
        if self.process_group is None:
            return extractFeatures(selected_images, self.img_height, self.img_width, self.model_online, 500, gpu_index=self.gpu_indexes[0],
                                   keep_on_device=True)
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
        if not parallel.buffers_in_sync((self.model_online,), self.process_group):
            raise _lib.DaliError("data parallel: BatchNorm running statistics differ between ranks before the epoch inference "
                                 "(parallel.sync_buffers_from_rank0 runs at the end of trainer.train; load the same checkpoint on every rank)")
        bounds = parallel.slice_bounds(len(selected_images), world)
        lo, hi = bounds[rank], bounds[rank + 1]
        local = extractFeatures(selected_images[lo:hi], self.img_height, self.img_width, self.model_online, 500, gpu_index=self.gpu_indexes[0],
                                keep_on_device=True)
        if local.shape[0] == 0:
            local = local.new_zeros(0, getattr(self._net, "feat_dim", None) or self._net.in_planes)
        self.last_inference_rows = hi - lo
        return parallel.all_gather_rows(local.float().contiguous(), bounds, self.process_group)

    def build_targets(self, selected_images, selected_labels):
        self.model_online.eval()
        print("Number of samples for proxies generation: %d" % selected_images.shape[0])
        fvs = self.extract_train_features(selected_images)
        centers, centers_labels, all_proxies, proxies_labels, mean_max = build_centers_and_proxies(fvs, selected_labels, self.num_proxies,
                                                                                                     self.process_group)
        pd = ops_eval.pairdist(all_proxies, all_proxies, metric="l2sq").clamp_(min=0).sqrt_()      # the cdist statistic, :148-153
        same = torch.from_numpy(proxies_labels[:, None] == proxies_labels[None, :]).to(pd.device)
        min_distance = torch.where(same, pd.max(), pd).min().item()
        print("Mean Max Proxies Positive Distances: %.3f, Min Negative Distance: %.3f" % (mean_max, min_distance))
        return LossHeads(centers, centers_labels, all_proxies, proxies_labels, self.tau, self.lambda_proxy, self.process_group)

    # ---- one optimisation step on a device-resident batch (train_encodersKIT.py:191-231) ----
    def train_step(self, heads, batch_imgs, labels_codes, w, acc):
        """acc: device fp32 [6] running sums (center, proxy, total, weights_sum, steps, _)."""
        net = self._net
        emb = net._run_forward(batch_imgs, training=True)
        fn = ops_eval.l2norm_rows(emb, 1e-9)                                          # :198
        stats, dfn = heads(fn, labels_codes, w)                                       # :200-208 (+ gradient)
        d_emb = ops_eval.l2norm_rows_bwd(emb, dfn, 1e-9)
        n_stages = net.n_bwd_stages
        if self.process_group is not None and self._dp is None:
            self._dp = parallel.GradReducer(net.flat_grads, [net._bwd_plan.stage_range(s) for s in range(n_stages)], self.process_group)
        for stage in range(n_stages):                                                 # :215 backward
            net._backward_stage(d_emb, stage)
            if self._dp is not None:
                self._dp.reduce_stage(stage)
        if self._dp is not None:
            self._dp.finish()
        self._adam.step()                                                             # :216
        optim.ema_update(self._mom, net, self.beta)                                   # :218-226
        total, lc, lp = LossHeads.losses_from_stats(stats, self.lambda_proxy)
        acc[0] += lc; acc[1] += lp; acc[2] += total
        acc[3:4] += self._adam.weights_sqsum                                          # :229-231
        acc[4] += 1
        return stats

    def train(self, selected_images, selected_labels, number_of_iterations, current_epoch):
        dev = torch.device("cuda", self.gpu_indexes[0])
        event_dataset = samplePKBatches(self.dataset, selected_images, selected_labels, self.img_height, self.img_width,
                                        self.turbulance_dir_path, self.kind_of_transform, K=self.K)
        num_classes = np.unique(selected_labels).shape[0]
        bs = min(self.P, num_classes)
        # data parallel (one process per GPU, replaces nn.DataParallel's scatter, Encoders.py:39-40): every rank walks rank 0's
        # identity order and takes its contiguous share of the P identities of each batch (parallel.shard_identities)
        world = rank = 0
        if self.process_group is not None:
            import torch.distributed as dist
            world, rank = dist.get_world_size(self.process_group), dist.get_rank(self.process_group)
            if bs % world != 0 or bs < 3:
                raise _lib.DaliError("data parallel: P=%d identities per batch must be >= 3 and divide over %d ranks" % (bs, world))
            event_dataset.labels_set = parallel.broadcast_from_rank0(event_dataset.labels_set, self.process_group)
        heads = self.last_targets = self.build_targets(selected_images, selected_labels)
        self.model_online.train()
        self.model_momentum.eval()
        for inner_iter in np.arange(number_of_iterations):
            print("Iteration number: %d/%d" % (inner_iter + 1, number_of_iterations))
            order = np.random.permutation(len(event_dataset))                         # DataLoader(shuffle=True, drop_last=True)
            if world:
                order = parallel.broadcast_from_rank0(order, self.process_group)
            n_batches = len(order) // bs
            acc = torch.zeros(6, device=dev, dtype=torch.float32)
            def batch_ids(b):
                ids = order[b * bs:(b + 1) * bs]
                return parallel.shard_identities(ids, rank, world) if world else ids
            # Loaders with the batched protocol (transforms.gpu_train_loader): batch b + 1 and b + 2 are planned (their random draws made, in
            # batch order) and decoding on the pool while step b is enqueued and runs; one resize + one augment launch per batch on a side
            # stream.  Other loaders (synthetic tensors in memory) keep the per-identity calls.
            loader = get_train_loader()
            batched = all(hasattr(loader, a) for a in ("plan", "submit", "finish"))
            pending = []
            nxt = 0
            for b in range(n_batches):
                if batched:
                    while nxt < n_batches and len(pending) < 1 + self.prefetch_depth:
                        pending.append(event_dataset.plan_batch(batch_ids(nxt), loader)); nxt += 1
                    batch_imgs, labels_f, dist_np = event_dataset.finish_batch(pending.pop(0), loader, dev)
                else:
                    parts = [event_dataset[i] for i in batch_ids(b)]
                    batch_imgs = torch.cat([p[0] for p in parts], 0).to(dev, non_blocking=True)
                    labels_f, dist_np = torch.cat([p[1] for p in parts], 0), np.concatenate([p[2] for p in parts])
                if not world and batch_imgs.shape[0] <= 2:                            # :194-195 (never under DP: the global batch has
                    continue                                                          # >= 3 identities, and a skip must be collective)
                labels_codes = _codes(labels_f, dev)
                w = _sample_weights(torch.from_numpy(dist_np), current_epoch, self.number_of_epoches, dev)
                self.train_step(heads, batch_imgs, labels_codes, w, acc)
            a = acc.cpu().numpy()                                                      # the only host sync of the epoch
            nb = max(n_batches, 1)
            self.last_epoch_stats = dict(center=a[0] / nb, proxy=a[1] / nb, loss=a[2] / nb, weights_sum=a[3] / nb, steps=int(a[4]))
            print("Batches computed: %d" % int(a[4]))
            print("Mean Center Loss: %.7f, Mean Proxy Loss: %.7f" % (a[0] / nb, a[1] / nb))
            print("Mean Final Loss: %.7f" % (a[2] / nb))
            print("Mean Weights Sum: %.2f" % (a[3] / nb))
        # nn.DataParallel keeps replica 0's BatchNorm running statistics (Encoders.py:39-40): every rank takes rank 0's, for the online net
        # and for the momentum net (whose buffers are the EMA of rank 0's), before anything runs in eval mode
        parallel.sync_buffers_from_rank0((self._net, self._mom), self.process_group)
        self.model_online.eval()
        self.model_momentum.eval()
