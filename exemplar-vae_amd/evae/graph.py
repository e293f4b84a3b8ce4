"""hipGraph capture of one whole training step (reference utils/training.py:27-46: binarise, loss, backward,
optimizer) so that a step costs ONE graph launch on the host instead of ~90 kernel launches plus the Python
between them.  At the 25 000-exemplar configuration a step is < 2 ms of GPU work; with the exemplars sharded
over 8 GPUs it is a few hundred microseconds, far below what eager launching can feed.

What varies between steps lives in ONE static int64 control block that is refreshed before each replay: the exemplar
indices (still drawn by the CPU generator exactly like reference models/BaseModel.py:245), the dataset indices of the
batch, beta and AdamNormGrad's bias-corrected step sizes.  The host writes a pinned copy (double-buffered), a copy
stream uploads it ahead of time into one of two device staging blocks, and the step takes it from there: on the byte store
its first launch copies the staging block into the control block itself (r06; a device-to-device copy in front of the graph
otherwise); thin steps upload straight into the block.
The batch images themselves are rows of the HBM-resident dataset: the graph gathers (and binarises) them by index, so
no image bytes cross PCIe (checked once against the loader's first batch; a loader that hands out other images gets
them uploaded instead).  The gather, the dynamic binarisation and the eps of the fused `vae` step are one launch of a
counter-based generator (evae_batch_prologue: seed = torch's seed when the runner is built, counter = step number, both
in the control block); whatever else draws random numbers uses the CUDA generator, which torch registers with the
capture so that every replay advances its Philox offset.  RCCL collectives of the sharded prior are captured too.
"""
import ctypes as C
import math
import os
import time
import sys

import torch

from . import _lib, ops, shard


_REFRESH_PROF = os.environ.get("EVAE_REFRESH_PROF") == "1"       # host time of _refresh by part, printed at exit (tools/host_time.py)


class _RefreshProf:
    acc = {}

    @staticmethod
    def lap(name, t0):
        t1 = time.perf_counter()
        a = _RefreshProf.acc.setdefault(name, [0.0, 0])
        a[0] += t1 - t0; a[1] += 1
        return t1


if _REFRESH_PROF:
    import atexit
    atexit.register(lambda: print("refresh: " + "; ".join("%s %.1f us" % (k, 1e6 * v[0] / max(v[1], 1)) for k, v in _RefreshProf.acc.items()),
                                  file=sys.stderr))


class GraphedTrainStep:
    def __init__(self, model, optimizer, dataset, batch_size, dynamic_binarization, warmup_steps=3):
        a = model.args
        self.model, self.opt, self.dataset = model, optimizer, dataset
        self.B = int(batch_size)
        self.binarize = bool(dynamic_binarization)
        dev = torch.device(a.device)
        D = int(torch.tensor(a.input_size).prod().item())
        C = int(a.number_components)
        self.x_in = torch.zeros((self.B, D), device=dev)
        # gather list of the fused step: [this rank's exemplar indices | the staging rows of the batch]; only the head
        # changes between steps, so the captured graph needs no arange/cat
        self.lo, self.hi = shard.bounds(C) if model._sharded() else (0, C)
        Cl = self.hi - self.lo
        # the fused `vae` step gathers its rows from the uint8 store when the data allow it (models/BaseModel.py::resident_u8)
        u8 = model.resident_u8(dataset, self.B) if (a.model_name == 'vae' and model._fused_config()) else None
        self.u8 = u8 is not None
        if self.u8:
            self.data_ext, self.n_data, self.x_div = u8
        else:
            self.data_ext, self.n_data = model.resident_data_ext(dataset, self.B)
        self.data_rows = self.data_ext[:self.n_data]
        self.stage_rows = self.data_ext[self.n_data:self.n_data + self.B]      # where the fused step wants the batch
        # Everything that varies between steps is ONE int64 control block in static device memory:
        #   [exemplar rows (Cl) | staging rows (B, constant) | batch dataset indices (B) | generator seed, step counter |
        #    beta, Adam step sizes (fp32)]
        # The host fills a pinned copy (double-buffered), a copy stream uploads it ahead of time into a device staging
        # block, and the step's own stream only takes it from there (the first launch's hand-over below, or one device-to-device
        # copy in front of the graph launch) -- it never waits for a DMA engine or for the host.
        self.ngroups = len(optimizer.param_groups)
        nsc = 1 + self.ngroups
        # Duplicates among the draw (the reference draws WITH replacement, models/BaseModel.py:245): when they are worth it the
        # gather list holds the DISTINCT rows only -- `cap` of them, a fixed count the distinct rows of a draw stay under by
        # five standard deviations (eight until r06), padded with multiplicity 0 -- and the block carries the draw itself, every draw's
        # position among the distinct rows, one draw per distinct row and the multiplicities behind its scalars
        # (evae_host_dedup on the host; evae/fused_vae.py::DEDUP for what the step does with them).  EVAE_DEDUP=0: off.
        self.dedup = None
        Cd = Cl                                  # rows of the gather list's head
        Nt = int(a.training_set_size)
        from . import fused_vae as _fv0
        sharded = model._sharded()
        want = (os.environ.get("EVAE_DEDUP", "1") != "0" and (Cl == C or sharded) and Cl > 0 and not a.approximate_prior and Nt > 0
                and a.prior == 'exemplar_prior' and int(a.z1_size) % 4 == 0)
        if want and a.model_name == 'vae' and model._fused_config() and not sharded:
            # the fused node on one device: what it does with the tables needs the byte store and the one-launch prior of a captured
            # step (the same predicate as there).  A sharded fused step gathers the per-draw centres for the prior's own kernels (r06);
            # every other model meets the tables in get_exemplar_set (models/BaseModel.py, ops.ExpandRowsFn)
            want = (self.u8 and os.environ.get("EVAE_UNIT_UPSTREAM", "1") != "0" and _fv0.PRIOR_TRAIN and not _fv0.ONE_STREAM[0]
                    and ops.prior_train_applies(self.B, C, int(a.z1_size)))
        if want:
            # distinct rows among the Cl draws of this process (all C on one device, its shard of the common draw otherwise) from Nt:
            # mean Nt (1 - q), variance Nt q (1 - q) + Nt (Nt - 1) (q2 - q^2) with q = (1 - 1/Nt)^Cl the chance that a given row is not
            # drawn, q2 = (1 - 2/Nt)^Cl that two given rows are not (the occupancies are negatively correlated: c2's 25 000 of
            # 50 000 give 19 673 +- 52); cap = mean + 5 sigma (+ 32 rows), whole 128-row tiles
            lq = Cl * math.log1p(-1.0 / Nt) if Nt > 1 else -math.inf
            q = math.exp(lq)
            dq = q * q * math.expm1(Cl * math.log1p(-2.0 / Nt) - 2.0 * lq) if Nt > 2 else 0.0       # q2 - q^2
            mean_u = Nt * (1.0 - q)
            var_u = max(Nt * q * (1.0 - q) + Nt * (Nt - 1.0) * dq, 1.0)
            # (r06: five standard deviations, it was eight -- a draw that does not fit is stepped eagerly once since r06, so the margin
            #  only has to make that rare: c2's 19 968 rows sit 5.6 sigma above the mean, ~1e-8 per step; 20 224 before)
            nsig = float(os.environ.get("EVAE_DEDUP_SIGMAS", "5"))
            cap = min(Cl, int(math.ceil((mean_u + nsig * math.sqrt(var_u) + 32.0) / 128.0)) * 128)
            if cap <= 0.92 * Cl:
                self.dedup = {"cap": cap, "distinct": 0}
                Cd = cap
        self._Cd = Cd
        self._o_idx, self._o_seed = Cd + self.B, Cd + 2 * self.B
        self._o_scal = self._o_seed + 2
        words = self._o_scal + (nsc + 1) // 2
        if self.dedup is not None:
            self._o_draw = words
            self._o_inv = self._o_draw + Cl
            self._o_rep = self._o_inv + Cl
            self._o_mult = self._o_rep + Cd
            words = self._o_mult + (Cd + 1) // 2
        words += words & 1                          # (whole 16-byte words: the prologue's hand-over copies uint4s)
        self.ctl = torch.zeros(words, dtype=torch.int64, device=dev)
        self.rows = self.ctl[:self._o_idx]
        self.idx_flat = self.ctl[self._o_idx:self._o_seed]
        self.seed_ctr = self.ctl[self._o_seed:self._o_scal]
        self.idx_in = self.idx_flat.view(self.B, 1)
        self.scal = self.ctl[self._o_scal:].view(torch.float32)[:nsc]
        self.beta = self.scal[0:1].reshape(())
        tail = torch.arange(self.n_data, self.n_data + self.B)
        self.rows[Cd:] = tail.to(dev)
        self._h_ctl = [torch.zeros(words, dtype=torch.int64).pin_memory() for _ in range(2)]
        for h in self._h_ctl:
            h[Cd:self._o_idx] = tail
        if self.dedup is not None:
            self.dedup["tables"] = (self.ctl[self._o_draw:self._o_inv], self.ctl[self._o_inv:self._o_rep],
                                    self.ctl[self._o_rep:self._o_mult], self.ctl[self._o_mult:].view(torch.float32)[:Cd])
            self.dedup["host_mult"] = [h[self._o_mult:].view(torch.float32) for h in self._h_ctl]
        # numpy views of the pinned blocks: a field write through them costs ~0.3 us, through a tensor index ~3 us (r03: the
        # replayed step of a small exemplar set is bound by the host -- 56 graph nodes at ~3.9 us of hipGraphLaunch each plus
        # this function -- so its microseconds are the step's)
        self._h_np = [h.numpy() for h in self._h_ctl]
        self._h_np_f32 = [h[self._o_scal:].view(torch.float32).numpy() for h in self._h_ctl]
        # Thin steps (host-bound): ONE upload on the step's stream straight into the control block instead of upload stream +
        # staging block + device copy (seven stream / event calls, 65 us of host time measured); the DMA's ~10 us then sit in
        # front of the graph, which a GPU-bound step (c2) would feel and a host-bound one does not.  EVAE_CTL_DIRECT=0/1 forces.
        # (r03 A/Bs, profiles/r03_ab/knobs.jsonl and tools/host_time.py: run-to-run spread at these sizes is +-10 %, the two paths
        # tie at C = 200 (0.25-0.28 either way) and the direct one is ahead at c1 (0.267 / 0.267 against 0.273 / 0.294).  The
        # staged path's six stream / event / copy operations go through ONE C call, evae_ctl_upload: 56 us of host time instead
        # of 65 -- the HIP calls themselves cost that, not the bindings.)
        e = os.environ.get("EVAE_CTL_DIRECT")
        self._direct = (Cd + self.B <= 8192) if e is None else e == "1"
        self._h_draw = torch.zeros(C, dtype=torch.int64)                 # the full draw when only a shard is uploaded
        self._d_ctl = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(2)]
        self._up = torch.cuda.Stream(device=dev)
        self._ev_up = [torch.cuda.Event() for _ in range(2)]             # upload k finished (host buffer k reusable)
        self._ev_used = [torch.cuda.Event() for _ in range(2)]           # device staging k consumed by the step stream
        for ev in self._ev_up + self._ev_used:
            ev.record()                            # (their handles exist from here on)
        self._ctl_bytes = words * 8
        # r06: on the byte store the step's FIRST launch hands the block over itself (evae_batch_prologue_u8_step: it reads the
        # batch indices and the counter from staging block `parity`, copies that block into self.ctl for every later launch, and
        # the step's LAST launch -- the optimizer's, which carries the statistics too -- flips the parity): no device-to-device
        # copy node in front of the graph.  The parity lives on the device; eager steps set it from the host's count.
        self._handover = ((not self._direct) and self.u8 and os.environ.get("EVAE_CTL_HANDOVER", "1") != "0"
                          and os.environ.get("EVAE_TAIL_MERGE", "1") != "0")
        self._ho_state = torch.zeros(2, dtype=torch.int32, device=dev)
        self._ho_par = [torch.full((1,), k, dtype=torch.int32, device=dev) for k in range(2)]
        self.by_index = None      # True once the loader's images are known to be rows of the resident dataset
        # by-index batches are gathered, binarised and given their eps by ONE launch with a counter-based generator
        # (evae_batch_prologue); the seed is torch's at construction time, the counter is the step number
        self.seed = int(torch.initial_seed())
        if model._sharded() and getattr(a, 'shard_batch', False):
            # data-parallel batches: every rank needs its own noise (replicated batches need the SAME noise on every rank)
            self.seed += 0x9E3779B97F4A7C15 * (shard.world()[0] + 1)
        self.seed &= 0x7FFFFFFFFFFFFFFF
        # ... unless the model carries its own noise hook (an instance-level _draw_eps: tests, deterministic runs)
        own_noise = a.model_name == 'vae' and '_draw_eps' not in model.__dict__
        self.eps_buf = torch.zeros((self.B, int(a.z1_size)), device=dev) if own_noise else None
        self._one = torch.ones((), device=dev)
        self.out = torch.zeros(3, device=dev)       # (loss, -RE, KL) of the last step
        self.totals = torch.zeros(3, device=dev)    # running sums since reset_totals()
        self._adam_tables = {}     # this graph's AdamNormGrad pointer tables and member lists (utils/optimizer.py)
        self.cache = None          # approximate prior: the latent cache in static buffers (set_cache)
        self.graph = None
        self.failed = False
        self._overflow = None      # set by _refresh when a draw has more distinct rows than the captured step holds: (draws | staging rows)
        self.overflow_steps = 0
        self.eager_opt = False     # True: the participants' step counts differ (resumed checkpoint): eager optimizer steps only
        self.warmup_steps = max(2, warmup_steps)    # call 0 learns the optimizer's participants, call 1 warms the captured form
        self._calls = 0

    def reset_totals(self):
        self.totals.zero_()

    def set_cache(self, cache):
        """Approximate prior (models/BaseModel.py:256-271): the per-epoch latent cache of utils.training.train_one_epoch.  The
        captured launches read and refresh it in place, so it lives in static buffers that every epoch's cache is copied into."""
        if self.cache is None or tuple(self.cache[0].shape) != tuple(cache[0].shape):
            assert self.graph is None, "the latent cache changed shape after the step was captured"
            self.cache = tuple(t.detach().clone() for t in cache)
        else:
            for dst, src in zip(self.cache, cache):
                dst.copy_(src.detach())
        return self.cache

    def _first_layer_split(self):
        """(w1h, w1g, prepared buffer) when the fused step on the byte store will run: its weight split then rides in the
        prologue's launch (one launch at the head of the step instead of two); None otherwise."""
        from . import fused_vae
        m = self.model
        if os.environ.get("EVAE_HEAD_MERGE", "1") == "0" or not (m._fused_config() and not m._sharded()):
            return None
        named = dict(m.named_parameters())
        wh, wg = named.get("q_z_layers.0.h.weight"), named.get("q_z_layers.0.g.weight")
        if wh is None or wg is None or not (wh.is_contiguous() and wg.is_contiguous() and wh.dtype == torch.float32):
            return None
        lib = fused_vae._lib.load()
        H, D = wh.shape
        prep = ops._workspace("u8prep", lib.evae_dense_u8_prepared_bytes(H, D), wh.device)
        fused_vae.PREP_DONE[prep.data_ptr()] = (wh.data_ptr(), wg.data_ptr())
        # the transposed weights the two large data gradients of the backward pass read on the split-bf16 kernel
        jobs = []
        Cl = getattr(self, "_Cd", self.hi - self.lo)      # the rows the fused node encodes (the distinct ones of a draw when dedup is on)
        wm, w2h, w2g = named.get("q_z_mean.weight"), named.get("q_z_layers.1.h.weight"), named.get("q_z_layers.1.g.weight")
        if (os.environ.get("EVAE_WT_HEAD", "1") != "0" and wm is not None and w2h is not None and w2g is not None
                and not m.args.approximate_prior and lib.evae_gemm_x6_applies(Cl, H, 0)):
            if getattr(self, "_wt_bufs", None) is None:       # this runner's own buffers: written in the head launch, read in the backward
                nb1 = lib.evae_dense_bwd_data_wt_bytes(wm.shape[0], wm.shape[1], 1)
                nb2 = lib.evae_dense_bwd_data_wt_bytes(w2h.shape[0], w2h.shape[1], 2)
                self._wt_bufs = (torch.empty(nb1, dtype=torch.uint8, device=wh.device), torch.empty(nb2, dtype=torch.uint8, device=wh.device))
            jobs = [(wm.detach(), None, self._wt_bufs[0]), (w2h.detach(), w2g.detach(), self._wt_bufs[1])]
            fused_vae.WT_DONE[(wm.data_ptr(), w2h.data_ptr(), w2g.data_ptr())] = self._wt_bufs
        # layer 2's weight images of the pre-split GEMMs (fused_vae's p6 path, when the last step took it): built by the head
        # launch too -- they depend on the weights alone -- instead of two launches on the side stream and a join in front of
        # layer 2's forward GEMM (rides with the control block's hand-over: evae_batch_prologue_u8_step)
        packs = []
        last = fused_vae.P6_LAST[0]
        if (os.environ.get("EVAE_P6_HEAD", "1") != "0" and self._handover and self.by_index and last is not None
                and w2h is not None and w2g is not None and last[:2] == (w2h.data_ptr(), w2g.data_ptr())):
            H2, w2_img, w2t_img = last[2], last[3], last[4]
            packs = [(0, w2h.detach(), w2g.detach(), H2, H2, H2, 1, 0, w2_img),
                     (1, w2h.detach(), w2g.detach(), H2, H2, H2, -1, lib.evae_p6_nks(2 * H2), w2t_img)]
            fused_vae.P6_DONE[(w2h.data_ptr(), w2g.data_ptr())] = True
        return wh.detach(), wg.detach(), prep, jobs, packs

    # the body that gets captured
    def _eager_tables(self):
        """The optimizer tables of this runner's captured-FORM steps that are issued eagerly (step_eagerly, _every_draw_step): the
        graph's step-size scalars, but their own pointer tables -- an eager step's gradient buffers are fresh allocations, and
        writing their addresses into the table the captured launches read would redirect every later replay."""
        t = getattr(self, "_adam_tables_eager", None)
        if t is None:
            t = self._adam_tables_eager = {}
        if "step_size" in self._adam_tables:
            t["step_size"] = self._adam_tables["step_size"]
        return t

    def _body(self, eager_opt=False, tables=None):
        if self.by_index:
            # the batch is rows `idx` of the HBM-resident dataset: gathered (and binarised) straight into the staging rows
            # of the fused step, no image bytes cross PCIe
            if self.u8:
                x = self.x_in
                job = None
                if self._handover:
                    job = (self._d_ctl[0], self._d_ctl[1], self.ctl, self._ho_state, self._o_idx, self._o_seed)
                    if not torch.cuda.is_current_stream_capturing():
                        self._ho_state[:1].copy_(self._ho_par[self._calls & 1])     # (eager: the block this call uploaded)
                prep = self._first_layer_split()
                ops.batch_prologue_u8(self.data_rows, self.idx_flat, self.binarize, self.seed_ctr, self.x_div, x,
                                      self.stage_rows, self.eps_buf, prepare=prep, ctl_job=job,
                                      packs=prep[4] if (prep is not None and job is not None) else None)
            else:
                x = self.stage_rows
                ops.batch_prologue(self.data_rows, self.idx_flat, self.binarize, self.seed_ctr, x, self.eps_buf)
        else:
            x = torch.bernoulli(self.x_in) if self.binarize else self.x_in
        self.opt.zero_grad(set_to_none=True)      # backward then installs the fused node's gradient buffers
        from . import fused_vae as _fv
        _fv.UNIT_UPSTREAM[0] = os.environ.get("EVAE_UNIT_UPSTREAM", "1") != "0"     # the backward below is loss.backward(ones), nothing else
        ops.STEP_BETA[0] = self.beta if _fv.UNIT_UPSTREAM[0] else None              # (the modular paths' prior: ops.prior_logp)
        _fv.DEDUP[0] = self.dedup["tables"] if self.dedup is not None else None
        try:
            loss, RE, KL = self.model.calculate_loss((x, self.idx_in), self.beta, average=True, dataset=self.dataset,
                                                     cache=self.cache)
        finally:
            _fv.UNIT_UPSTREAM[0] = False
            _fv.DEDUP[0] = None
            ops.STEP_BETA[0] = None
        if self.by_index and self.u8:
            from . import fused_vae
            fused_vae.PREP_DONE.clear()           # (a token the fused step did not consume must not outlive this step)
            fused_vae.WT_DONE.clear()
            fused_vae.P6_DONE.clear()
        with ops.deferred_wgrads(loss):          # thin layers' weight gradients behind the backward pass, grouped (evae/ops.py)
            loss.backward(gradient=self._one)
        # the step's statistics ride in the optimizer's last launch (evae_adam_normgrad_step_stats)
        stats = (loss.detach(), RE.detach(), KL.detach(), self.out, self.totals) if os.environ.get("EVAE_TAIL_MERGE", "1") != "0" else None
        ho = self._handover and self.by_index
        if ho and stats is not None:
            stats = stats + (self._ho_state,)     # (the parity of the control block's staging blocks: flipped by the same launch)
        if eager_opt:
            # the reference's own bookkeeping (a step count per parameter, host-side step size): the runner's first call, which
            # learns which parameters take part at all, and every call of a runner whose participants disagree on the count
            self.opt.step()
            self.opt._stats_done = False
        else:
            self.opt.step(_captured=True, _tables=self._adam_tables if tables is None else tables, _stats=stats)
        if not getattr(self.opt, "_stats_done", False):
            ops.step_stats_add(loss.detach(), RE.detach(), KL.detach(), self.out, self.totals)
            if ho:
                self._ho_state[:1].bitwise_xor_(1)
        return self.out

    def _refresh(self, data, indices, beta):
        k = self._k_used = self._calls & 1
        Cl = self.hi - self.lo
        a = self.model.args
        if self.by_index is None:                 # once: are the loader's images the resident rows its indices name?
            ii = indices.reshape(-1).to(self.ctl.device)
            ok = bool(ii.numel() == self.B and int(ii.min()) >= 0 and int(ii.max()) < self.n_data)
            rows_f = self.data_rows.index_select(0, ii)
            if self.u8:
                rows_f = rows_f.float() / self.x_div
            xf = data.reshape(self.B, -1).to(self.ctl.device, torch.float32)
            self.by_index = ok and torch.equal(rows_f, xf)
            if not self.by_index and self.u8 and not torch.equal(torch.round(xf * self.x_div).clamp_(0, 255) / self.x_div, xf):
                # images that are neither rows of the dataset nor k/255: the byte store's staging rows cannot hold them.  The
                # eager step checks every batch and takes the fp32 store when it must (models/BaseModel.py); no capture
                print("evae.graph: the loader's images are not k/255 values; the step is not captured", file=sys.stderr)
                self.failed = True
        T = _RefreshProf if _REFRESH_PROF else None
        t0 = time.perf_counter() if T else 0.0
        self._ev_up[k].synchronize()              # the upload issued two steps ago from this host buffer is done
        if T: t0 = T.lap("wait for the block's last upload", t0)
        h = self._h_ctl[k]
        # same CPU-generator draw, with replacement, as the reference (models/BaseModel.py:245)
        if self.dedup is not None:
            dr = h[self._o_draw:self._o_inv]
            if Cl == a.number_components:
                torch.randint(low=0, high=a.training_set_size, size=(Cl,), out=dr)
            else:                                 # sharded: the common draw (same CPU generator state on every rank), this rank's slice
                torch.randint(low=0, high=a.training_set_size, size=(a.number_components,), out=self._h_draw)
                dr.copy_(self._h_draw[self.lo:self.hi])
            cap = self.dedup["cap"]
            nu = _lib.load().evae_host_dedup(C.c_void_p(dr.data_ptr()), Cl, int(a.training_set_size), cap, C.c_void_p(h.data_ptr()),
                                             C.c_void_p(h[self._o_inv:].data_ptr()), C.c_void_p(h[self._o_rep:].data_ptr()),
                                             C.c_void_p(self.dedup["host_mult"][k].data_ptr()))
            if nu < 0:
                # more distinct rows than the captured step has room for (its fixed count sits five standard deviations above the
                # mean: ~1e-15 per draw): THIS step is issued eagerly with every draw encoded -- the same loss and gradients
                # (__call__ / step_eagerly look at _overflow) -- and the next one replays again
                Cd_ = self._Cd
                self._overflow = torch.cat((dr, h[Cd_:self._o_idx])).to(self.ctl.device)
                h[:Cd_] = dr[:Cd_]                # (a valid gather list for the block that is uploaded all the same; never read)
                nu = 0
            self.dedup["distinct"] = nu
        elif Cl == a.number_components:
            torch.randint(low=0, high=a.training_set_size, size=(Cl,), out=h[:Cl])
        else:
            torch.randint(low=0, high=a.training_set_size, size=(a.number_components,), out=self._h_draw)
            h[:Cl] = self._h_draw[self.lo:self.hi]
        if T: t0 = T.lap("candidate draw", t0)
        idx_on_device = indices.is_cuda
        hn = self._h_np[k]
        if not idx_on_device:
            hn[self._o_idx:self._o_seed] = indices.reshape(-1).numpy()
        hn[self._o_seed] = self.seed
        hn[self._o_seed + 1] = self._calls
        hs = self._h_np_f32[k]
        hs[0] = float(beta)
        if T: t0 = T.lap("control block fields", t0)
        if self._calls > 0 and not self.eager_opt:      # (call 0 steps eagerly and learns the participants)
            self.opt.advance_graph_step(host_out=hs[1:1 + self.ngroups], tables=self._adam_tables)
        if T: t0 = T.lap("optimizer step counts", t0)
        if self._direct:
            self.ctl.copy_(h, non_blocking=True)
            self._ev_up[k].record()               # (on the step's stream: host block k is free once this upload ran)
        else:
            # upload stream: wait until staging block k was consumed (two steps ago), copy the host block there; step stream: wait
            # for that upload, one device-to-device copy into the control block (evae_ctl_upload: the six operations as one call)
            ho = self._handover and self.by_index
            _lib.check(_lib.load().evae_ctl_upload(C.c_void_p(self._d_ctl[k].data_ptr()), C.c_void_p(h.data_ptr()),
                                                   None if ho else C.c_void_p(self.ctl.data_ptr()), self._ctl_bytes,
                                                   C.c_void_p(self._up.cuda_stream), C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                   C.c_void_p(self._ev_used[k].cuda_event), C.c_void_p(self._ev_up[k].cuda_event)),
                       "evae_ctl_upload")
        if idx_on_device:
            dst = self._d_ctl[k][self._o_idx:self._o_seed] if (self._handover and self.by_index and not self._direct) else self.idx_flat
            dst.copy_(indices.reshape(self.B))
        if not self.by_index:
            self.x_in.copy_(data.reshape(self.B, -1), non_blocking=True)
        if T: T.lap("upload + copies", t0)

    def step_eagerly(self, data, indices, beta):
        """The SAME step -- same control block, same launches, same shapes (the distinct exemplar rows, the captured optimizer
        form) -- issued launch by launch instead of replayed: bench.py brackets its large launches with HIP events this way (event
        pairs cannot be read back from inside a replayed graph).  Only after the graph exists."""
        assert self.graph is not None, "step_eagerly: the step has not been captured yet"
        self.model._exemplar_indices_override = (self.rows, self._Cd)
        self.model._exemplar_dedup = self.dedup["tables"] if self.dedup is not None else None
        try:
            self._refresh(data, indices, beta)
            self.model._eps_override = self.eps_buf if self.by_index else None
            self.model._batch_staged = bool(self.by_index and self.u8)
            if self._overflow is not None:
                return self._every_draw_step()
            self._body(eager_opt=self.eager_opt, tables=self._eager_tables())
            self._calls += 1
            return self.out
        finally:
            if self._handover and self.by_index:
                self._ev_used[self._k_used].record()      # (staging block k was read by this step's first launch)
            self.model._exemplar_indices_override = None
            self.model._exemplar_dedup = None
            self.model._eps_override = None
            self.model._batch_staged = False

    def _every_draw_step(self):
        """One step issued eagerly with EVERY draw encoded (no distinct-row tables): what a draw that does not fit the captured
        step's fixed row count gets instead of an exception.  Same control block (batch, seed, beta, step sizes), same optimizer form."""
        rows_ext, self._overflow = self._overflow, None
        Cl = self.hi - self.lo
        self.model._exemplar_indices_override = (rows_ext, Cl)
        self.model._exemplar_dedup = None
        dd, self.dedup = self.dedup, None
        try:
            self._body(eager_opt=self.eager_opt or self._calls == 0, tables=self._eager_tables() if self.graph is not None else None)
            if self._calls == 0 and not self.eager_opt and not self.opt.learn_members(self._adam_tables):
                self.failed = self.eager_opt = True
        finally:
            self.dedup = dd
        self._calls += 1
        self.overflow_steps += 1
        return self.out

    def __call__(self, data, indices, beta):
        """One training step; returns a device tensor (loss, -RE, KL) valid until the next call."""
        if self.graph is None and self._calls == 0:
            # this graph's launches read the step size from ITS control block (another runner on the same optimizer has its own)
            self._adam_tables["step_size"] = [self.scal[1 + g:2 + g] for g in range(self.ngroups)]
            self.opt.enable_graph_mode(storage=self._adam_tables["step_size"])
        self.model._exemplar_indices_override = (self.rows, self._Cd)
        self.model._exemplar_dedup = self.dedup["tables"] if self.dedup is not None else None
        try:
            self._refresh(data, indices, beta)
            self.model._eps_override = self.eps_buf if self.by_index else None
            self.model._batch_staged = bool(self.by_index and self.u8)
            if self._overflow is not None:
                return self._every_draw_step()
            if self.graph is None:
                # eager warm-up steps on a side stream (workspaces, attributes, RCCL channels), then capture
                if self._calls < self.warmup_steps and not self.failed:
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        out = self._body(eager_opt=self._calls == 0)
                    torch.cuda.current_stream().wait_stream(s)
                    if self._calls == 0 and not self.opt.learn_members(self._adam_tables):
                        print("evae.graph: the optimizer's parameters carry different step counts (a resumed checkpoint); "
                              "the step is not captured and AdamNormGrad steps eagerly", file=sys.stderr)
                        self.failed = self.eager_opt = True
                    self._calls += 1
                    return out
                if self.failed:
                    # capture was refused: keep stepping eagerly.  A refusal on the very first call (_refresh: images the byte
                    # store cannot hold) comes before the participants of the captured optimizer form are known and before
                    # advance_graph_step ever ran (the control block's step sizes are still 0): that call steps the reference's
                    # way and learns the members, as a warm-up call would
                    first = self._calls == 0
                    self._body(eager_opt=self.eager_opt or first)
                    if first and not self.eager_opt and not self.opt.learn_members(self._adam_tables):
                        self.eager_opt = True
                    self._calls += 1
                    return self.out
                torch.cuda.synchronize()
                main_stream = torch.cuda.current_stream()
                graph = torch.cuda.CUDAGraph()
                # with RCCL in the step its watchdog thread polls events while we capture: thread-local capture mode
                # keeps those calls from invalidating the capture
                mode = "thread_local" if shard.is_active() else "global"
                mode = os.environ.get("EVAE_CAPTURE_MODE", mode)
                try:
                    with torch.cuda.graph(graph, capture_error_mode=mode):
                        self._body()
                    self.opt.finish_capture(self._adam_tables)
                    self.graph = graph
                except Exception as e:           # an op of this model that cannot be captured: eager from here on
                    print("evae.graph: hipGraph capture of the training step failed (%s: %s); running eagerly"
                          % (type(e).__name__, str(e).splitlines()[0][:120]), file=sys.stderr)
                    self.failed = True
                    # torch.cuda.graph.__exit__ does not restore the stream when capture_end() itself raises: the current
                    # stream would stay the (invalidated, still "capturing") capture stream and every later launch fail
                    torch.cuda.set_stream(main_stream)
                    for _ in range(4):           # a failed capture can leave a sticky HIP error behind: drain it
                        try:
                            torch.cuda.synchronize()
                            break
                        except Exception:
                            pass
                    self._body()
                    self._calls += 1
                    return self.out
                # capture does not execute: replay once for this call's step
            self.graph.replay()
            self._calls += 1
            if shard.is_active() and self._calls % shard.REPLICA_CHECK_EVERY == 0:
                shard.check_replicas(self.model.parameters())      # (replica mode's guard: raises when the ranks drifted apart)
            return self.out
        finally:
            if self._handover and self.by_index:
                self._ev_used[self._k_used].record()      # (staging block k was read by this step's first launch)
            self.model._exemplar_indices_override = None
            self.model._exemplar_dedup = None
            self.model._eps_override = None
            self.model._batch_staged = False
