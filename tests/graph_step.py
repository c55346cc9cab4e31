"""A whole training step captured ONCE in a HIP graph and replayed -- TEST INFRASTRUCTURE, not a product path.

Rounds 3 - 4 shipped this as ``runner.trainer.GraphedTrainStep``.  It is correct (tests/test_gpu_graph.py) and it never won: replaying
the ~600-node graph costs the ROCm 7.2 graph executor about what the eager launches cost the host, on every size measured -- one
encoder at batch 4: 1.82 ms replayed against 1.41 eager (1.67 eager on the dense rows a graph is confined to); the four-level wrapper on
one stream at batch 1 / 2 / 4: 3.74 / 3.91 / 5.52 ms replayed against 3.65 / 3.78 / 5.01 eager, although the host enqueues a replay in
0.55 ms instead of 3.0 - 3.6 (``tools/wrapper_time.py`` with GRAPH=1, round 4).  So the class left the trainer's API and the bench's
legs; what stays in the product is what makes the runtime CAPTURABLE (the device step clock every dropout key and the optimiser's step
count read, ``TfRadamArgs.lr_dev``), and these tests keep that true.
"""
import torch

from transfusion_amd.runner.trainer import FusionTrainStep


def check_capturable(module, overlap_on: bool):
    """Raises ValueError for a module tree whose step cannot be captured in a HIP graph on this runtime: a wrapper whose last forward
    ran its feature levels on LEVEL STREAMS (``_last_path == "streams"``) while the encoders fork a side stream of their own for the weight
    gradients.  Every fork of that step is joined (an encoder call waits for its side stream's done event on the level stream it ran on;
    the wrapper makes the origin wait for every level stream), so the capture is legal by the API's rules and each fork level alone
    captures and replays correctly -- but ending a capture with the SECOND level of forks present crashes inside hipStreamEndCapture
    (ROCm 7.2: a segmentation fault in the runtime, gpurun_out/wg.txt of round 3, not an error code).  Refusing here turns a core dump
    into an exception that names the two ways out."""
    for m in module.modules():
        if getattr(m, "_last_path", None) == "streams" and overlap_on:
            raise ValueError(
                f"GraphedTrainStep: {type(m).__name__} runs its feature levels on their own streams and every level's encoder forks a side "
                "stream for its weight gradients; hipStreamEndCapture crashes on that nested fork (ROCm 7.2).  Capture with the levels on "
                "one stream (TF_LEVEL_STREAMS=0), without the side streams (TF_WGRAD_OVERLAP=0), or with the levels as one grouped call "
                "(parameters in FusionTrainStep's flat layout, equal token grids).")


class GraphedTrainStep:
    """A whole training step of a ``FusionTrainStep`` -- forward, backward, clip, fused RAdam -- captured ONCE in a HIP graph and
    replayed: one host call per step instead of several hundred kernel launches, event records and autograd nodes.  For the
    reference's own per-GPU batch (4 - 5 samples) the step of the four-level wrapper is bound by exactly that host work.

    What makes a replay a NEW step rather than a copy of the captured one:
      * the library's step clock (``ops.clock_*``, tf_clock_ptr): the captured sequence starts by advancing it, every dropout site
        folds it into its key, so every replay draws fresh masks (forward and backward of one replay agree);
      * the optimiser reads its step number from the same clock and forms RAdam's schedule terms on the device;
      * the batch lives in static tensors: ``load(batch)`` copies the next batch in, ``replay()`` runs the step.
    Restrictions: one GPU (no collectives inside the graph), one micro-batch per step, dense rows (packed batches size their grids
    from the batch's token count, which a graph cannot change), tensors of fixed shape.  The eager path stays the reference semantics;
    tests/test_gpu_graph.py checks that replays and eager steps on the same clock give the same parameters."""

    def __init__(self, trainer: FusionTrainStep, batch, loss_fn, warmup: int = 3):
        from transfusion_amd import ops
        if trainer.world != 1:
            raise ValueError("GraphedTrainStep: one GPU only (the gradient exchange is not captured)")
        if trainer.accumulate != 1:
            raise ValueError("GraphedTrainStep: one micro-batch per optimiser step")
        import inspect
        if "on_clock" not in inspect.signature(trainer.opt.step).parameters:
            raise ValueError("GraphedTrainStep needs an optimiser whose step number can live on the device (FusedRAdam)")
        self.trainer, self.batch, self.loss_fn, self._ops = trainer, batch, loss_fn, ops
        for m in trainer.module.modules():
            if hasattr(m, "pack_tokens"):
                m.pack_tokens = False
        ops.clock_ptr()                                  # allocate + publish the clock before anything is captured
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):             # lazily created state (function attributes, side streams, shadows) settles here
            for _ in range(max(1, warmup)):
                ops.clock_advance()
                trainer.step([batch], loss_fn, on_clock=True)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        check_capturable(trainer.module, ops.wgrad_overlap_enabled())
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            ops.clock_advance()
            self.loss = trainer.step([batch], loss_fn, on_clock=True)
        # capturing RECORDED one step, it did not run it: take its host-side bookkeeping back (replay() redoes it per replay)
        ops._clock_host[0] -= 1
        for st in trainer.opt.state.values():
            if "step" in st:
                st["step"] -= 1
        self.replays = 0

    def load(self, batch):
        """Copies ``batch`` (same tree of tensors, same shapes) into the static tensors the graph reads."""
        def copy(dst, src):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)
            elif isinstance(dst, dict):
                for k in dst:
                    copy(dst[k], src[k])
            elif isinstance(dst, (list, tuple)):
                for d, s_ in zip(dst, src):
                    copy(d, s_)
        copy(self.batch, batch)

    def replay(self):
        """One optimiser step.  Returns the (static) loss tensor of that step.  The learning rate of every parameter group is read from
        a device scalar that is refreshed here (an LR scheduler may have moved ``param_groups[i]['lr']`` since the last replay)."""
        if hasattr(self.trainer.opt, "refresh_lr"):
            self.trainer.opt.refresh_lr()                # (on the current stream: the one graph.replay() launches on)
        self.graph.replay()
        self.replays += 1
        self._ops._clock_host[0] += 1                    # the graph advanced the device word
        for st in self.trainer.opt.state.values():
            if "step" in st:
                st["step"] += 1
        return self.loss

    def finish(self):
        """Call before going back to eager calls on the module: the parameter versions the shadow caches key on."""
        self.trainer.mark_parameters_updated()
