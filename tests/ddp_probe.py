"""Shared by the world-size-2 trainer tests (gloo on CPU, gloo on one GPU): proves that an optimiser step of the
data-parallel trainer IS the step a single process would take with the mean of the shards' gradients.

Replica equality alone cannot show that (a SUM instead of a MEAN, a bucket counted twice or 1/world applied twice are
identical on every rank), so the local, pre-exchange gradient of every bucket is snapshotted at the moment the bucket
is handed to the collective, gathered over the ranks and averaged by the test itself."""
import torch
import torch.distributed as dist


class StepProbe:
    def __init__(self, trainer):
        self.trainer = trainer
        self.local = {}                      # label -> flat local gradient (bucket order) of that step
        self._pending = {}
        for red in (trainer.generator_reducer, trainer.discriminator_reducer):
            self._wrap_launch(red)
        orig_step = trainer._step

        def step(reducer, optimizer, label=""):
            orig_step(reducer, optimizer, label)         # finish() inside launches whatever the hooks did not
            got = self._pending.pop(id(reducer))
            assert sorted(got) == list(range(len(reducer.buckets))), "a bucket was never (or twice) exchanged"
            self.local[label] = torch.cat([got[i] for i in range(len(reducer.buckets))])
        trainer._step = step
        trainer.step_trace = {}

    def _wrap_launch(self, red):
        orig = red._launch

        def launch(bucket):
            idx = next(i for i, b in enumerate(red.buckets) if b is bucket)
            slot = self._pending.setdefault(id(red), {})
            assert idx not in slot, "bucket launched twice in one step"
            # (the parameters' slices, without the alignment padding between them)
            slot[idx] = torch.cat([bucket.flat[o:o + p.numel()].detach() for p, o in zip(bucket.params, bucket.offsets)])
            orig(bucket)
        red._launch = launch

    def check(self, world, lr=None, tol=1e-5):
        """trace gradient == mean over ranks of the local gradients (per label); with plain SGD of rate `lr` also
        movement == -lr * clip_coef * that mean."""
        tr = self.trainer
        for label, local in self.local.items():
            red = tr.discriminator_reducer if label in ("d", "r1") else tr.generator_reducer
            if dist.get_backend() != "nccl":             # (RCCL gathers device tensors; gloo takes either)
                local = local.cpu()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            mean = torch.stack(gathered).mean(0).cpu()
            names = [tr._param_names[id(p)] for b in red.buckets for p in b.params]
            got = torch.cat([tr.step_trace[f"{label}.grad.{n}"].flatten() for n in names]).cpu()
            scale = mean.abs().max().item()
            assert scale > 0 and (got - mean).abs().max().item() <= tol * scale, \
                (label, "exchanged gradient is not the mean of the shards", (got - mean).abs().max().item(), scale)
            norm = float(tr.step_trace[f"{label}.gnorm"])
            assert abs(norm - mean.norm().item()) <= 1e-4 * mean.norm().item(), (label, norm, mean.norm().item())
            if lr is not None:
                coef = min(1.0, 5.0 / (norm + 1e-6))
                delta = torch.cat([tr.step_trace[f"{label}.delta.{n}"].flatten() for n in names]).cpu()
                want = -lr * coef * mean
                assert (delta - want).abs().max().item() <= 1e-3 * want.abs().max().item() + 1e-9, \
                    (label, "parameter movement is not the SGD step of the mean gradient")
        return sorted(self.local)
