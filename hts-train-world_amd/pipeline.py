"""Host-to-host pipeline: waveforms in pinned host memory in, feature files' contents in pinned host memory out.

SURVEY.md 8(d) counts the metric "including H2D of waveforms and D2H of features", and 8(e) asks for double-buffered
pinned copies.  The types on the wire are the on-disk ones of the reference's CLIs:

    in   int16 samples (the wav payload; x = s / 32768 is formed on the device, test/audioio.cpp:236-249)   2 B/sample
    out  float32 f0 [T], sp [T][F/2+1], ap [T][F/2+1] (test/analysis.cpp:360-390)                            4 B/value
         -- or, coded=(spec_dim, ap_dim), what the recipe's own call `analysis ... 5 2048 50 25` writes
         (data/Makefile.in:214): float32 lf0 [T], mgc [T][50], bap [T][25] (analysis.cpp:292-366) --
         int16 resynthesised samples (test/audioio.cpp:160-167), when synthesis is on

Three HIP streams: uploads, kernels (the context's), downloads; two slots of device and pinned buffers.  While the
kernels of step k run, the waveforms of step k+1 travel up and the features of step k-1 travel down; PCIe and the
kernels overlap, so a step costs max(kernels, download) instead of their sum (the download dominates: 4.1 KB per frame
against 160 B up).

Order matters on the copy engine: copies of both directions are served in the order they were issued, so the (small)
upload of step k+1 has to be issued BEFORE the (large) download of step k, or the next step's kernels wait for that
whole download to drain.  Hence the two calls: feed() issues an upload, submit() the kernels and the download of the
oldest fed step; keep one upload ahead:

    feed()                       # step 0
    for k in range(K):
        fill input_buffer(); feed()      # step k + 1 (while there is one)
        slot = submit()                  # step k
        ... result(previous slot) ...

torch is plumbing here (streams, events, pinned memory, dtype conversion); the analysis is libworld_mi355.so's.
"""
from __future__ import annotations

import numpy as np

from . import world as W


class HostPipeline:
    """Fixed batch shape (a list of utterance lengths), processed repeatedly with different waveforms."""

    SLOTS = 2

    def __init__(self, ctx, params, x_lengths, synthesis=True, coded=None):
        import torch
        self.torch = torch
        self.batch = b = W.WorldBatch(ctx, params, x_lengths=x_lengths)
        self.synthesis = synthesis
        self.coded = coded
        T, bins, n, ny = int(b.total_frames), b.bins, int(b.total_samples), int(b.total_out)
        w_sp, w_ap = (coded if coded else (bins, bins))       # widths of the two downloaded feature arrays
        dev = lambda *shape, dtype: torch.empty(*shape, dtype=dtype, device="cuda")
        pin = lambda *shape, dtype: torch.empty(*shape, dtype=dtype, pin_memory=True)
        f32, f64, i16 = torch.float32, torch.float64, torch.int16
        self.x_pinned = [pin(n, dtype=i16) for _ in range(self.SLOTS)]
        self.x_dev = [dev(n, dtype=i16) for _ in range(self.SLOTS)]
        self.x64 = dev(n, dtype=f64)
        self.out64 = (dev(T, dtype=f64), dev(T, dtype=f64), dev(T, bins, dtype=f64), dev(T, bins, dtype=f64))
        self.y64 = dev(ny, dtype=f64)
        self.out32 = [(dev(T, dtype=f32), dev(T, w_sp, dtype=f32), dev(T, w_ap, dtype=f32), dev(ny, dtype=i16))
                      for _ in range(self.SLOTS)]
        self.host = [(pin(T, dtype=f32), pin(T, w_sp, dtype=f32), pin(T, w_ap, dtype=f32), pin(ny, dtype=i16))
                     for _ in range(self.SLOTS)]
        self.compute = torch.cuda.current_stream()
        self.up, self.down = torch.cuda.Stream(), torch.cuda.Stream()
        ev = lambda: [torch.cuda.Event() for _ in range(self.SLOTS)]
        self.ev_up, self.ev_x_free, self.ev_done, self.ev_down = ev(), ev(), ev(), ev()
        for s in range(self.SLOTS):                          # everything starts free
            self.ev_x_free[s].record(self.compute)
            self.ev_down[s].record(self.down)
        self.step = 0                                        # steps submitted
        self.fed = 0                                         # steps fed
        self.pending_down = None                             # slot whose download has not been issued yet
        # a short download (coded features) is issued late, beside the next step's CheapTrick / D4C (_issue_download);
        # a long one (raw float32 sp / ap: longer than a step on its own) at once: it spans the next step anyway
        self.defer_down = coded is not None

    def input_buffer(self):
        """The pinned int16 buffer of the next feed() (numpy view); fill it, then call feed()."""
        s = self.fed % self.SLOTS
        self.ev_up[s].synchronize()                          # the slot's previous upload has left the buffer
        return self.x_pinned[s].numpy()

    def feed(self):
        """Issue the upload of the buffer input_buffer() returned.  At most SLOTS steps may be fed and not submitted."""
        assert self.fed - self.step < self.SLOTS, "feed() is more than %d steps ahead of submit()" % self.SLOTS
        torch = self.torch
        s = self.fed % self.SLOTS
        with torch.cuda.stream(self.up):
            self.up.wait_event(self.ev_x_free[s])            # the kernels that read this slot last have consumed it
            self.x_dev[s].copy_(self.x_pinned[s], non_blocking=True)
            self.ev_up[s].record(self.up)
        self.fed += 1

    def submit(self):
        """Issue the kernels and the download of the oldest fed step (feeds it first if nothing is fed).  Returns
        the step's slot; everything is asynchronous except the one host round trip inside Synthesis."""
        torch = self.torch
        if self.fed == self.step:
            self.feed()
        s = self.step % self.SLOTS
        b = self.batch
        self.compute.wait_event(self.ev_up[s])
        x = b.samples_from_pcm16(self.x_dev[s], out=self.x64)              # wavread: s / 32768, one pass
        self.ev_x_free[s].record(self.compute)
        f0, sp, ap, y = self.out64[1], self.out64[2], self.out64[3], None
        if self.synthesis:
            b.analyze_synthesize(x, out=self.out64, y=self.y64)
            y = self.y64
        else:
            b.analyze(x, out=self.out64)
        self._issue_download()                                             # of the step before this one (see there)
        self.compute.wait_event(self.ev_down[s])                           # slot s's previous download has finished
        o = self.out32[s]
        if self.coded:
            lf0, mgc, bap = b.recipe_features(f0, sp, ap, self.coded[0], self.coded[1])
            o[0].copy_(lf0)
            o[1].copy_(mgc)
            o[2].copy_(bap)
        else:
            o[0].copy_(f0)
            o[1].copy_(sp)
            o[2].copy_(ap)
        if y is not None:                                                  # wavwrite: clamp(int(y * 32767)), one pass
            b.samples_to_pcm16(y, out=o[3])
        self.ev_done[s].record(self.compute)
        self.pending_down = s
        if not self.defer_down:
            self._issue_download()
        self.step += 1
        return s

    def _issue_download(self):
        """The download of the step submitted last.  It is issued one step LATE, right after the next step's kernels
        have been queued: that call returns when the device is past the next step's Dio and StoneMask (the host round
        trip inside Synthesis), so the copy -- a kernel on this platform, not a DMA transfer -- runs beside CheapTrick
        and D4C, which hardly touch memory.  Issued at once it ran beside the next step's Dio kernels, which are made
        of memory traffic and took 2.3 ms longer for it (dio_mean_partial 0.03 -> 0.93 ms, dio_band_compact 0.2 ->
        0.84 ms: profiles/r04_d_kernel_trace.csv)."""
        s = self.pending_down
        if s is None:
            return
        torch = self.torch
        with torch.cuda.stream(self.down):
            self.down.wait_event(self.ev_done[s])
            for h, d in zip(self.host[s], self.out32[s]):
                h.copy_(d, non_blocking=True)
            self.ev_down[s].record(self.down)
        self.pending_down = None

    def result(self, slot):
        """Wait for the download of `slot`; returns numpy views (f0, sp, ap, y_int16) -- (lf0, mgc, bap, y_int16) when
        coded -- of its pinned buffers, valid until that slot is submitted again."""
        if self.pending_down == slot:
            self._issue_download()
        self.ev_down[slot].synchronize()
        return tuple(h.numpy() for h in self.host[slot])

    def bytes_per_step(self):
        b = self.batch
        up = 2 * int(b.total_samples)
        widths = sum(self.coded) if self.coded else 2 * b.bins
        down = 4 * int(b.total_frames) * (1 + widths) + (2 * int(b.total_out) if self.synthesis else 0)
        return up, down

    def close(self):
        self._issue_download()
        self.torch.cuda.synchronize()
        self.batch.close()


def to_int16(x):
    """float samples s / 32768 (synth_data, wavread) back to the int16 payload."""
    return np.round(np.asarray(x) * 32768.0).astype(np.int16)
