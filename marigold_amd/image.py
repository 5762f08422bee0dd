"""Model images: the pipeline's native programs for ONE problem shape, compiled ahead of time into a single file that
``libmarigold_hip.so`` loads and runs WITHOUT Python (``mg_model_load`` / ``mg_model_vae_encode`` / ``mg_model_denoise`` /
``mg_model_vae_decode`` in include/marigold_hip.h) - the module-level C entry points SURVEY.md section 8(b) proposes for the
seams ``single_infer`` calls (marigold_depth_pipeline.py:396-477: ``encode_rgb`` :491-495, the T-step ``unet`` +
``scheduler.step`` loop :455-468, ``decode_depth`` :498-516).

The Python engine stays the program BUILDER (graph emission in engine.py, weight packing in weights.py run once, here, at
export time); what a host in any language needs at run time is the flat op lists, the kernel-ready weights and a memory plan -
exactly what this file holds:

    header | buffer table | program table (+ named input / output slots) | relocation table | op arrays | weight bytes

Every device pointer inside an ``mg_op`` is recorded as (buffer, byte offset) - the loader allocates the buffers, uploads the
weights, zeroes what must start zeroed and patches the pointers.  ``export_model_image`` works on a host without a GPU (the
programs are built against host buffers exactly as ``mg_program_validate`` does), so the CPU test-suite round-trips an image
through the C loader.
"""
import ctypes
import struct

import torch

from . import _lib as L
from . import ops as O
from .modules import AutoencoderKLHIP, UNet2DConditionModelHIP

MAGIC = b"MGIMG1\0\0"
VERSION = 1
KIND_SCRATCH, KIND_ZERO, KIND_DATA = 0, 1, 2
SLOT_LN_COUNTERS = 100   # relocation slot of the (i[29], i[30]) address pair of MG_OP_IGEMM; slots 0..15 = p[k]
_PRED = {"depth": (L.POST_DEPTH, 1), "normals": (L.POST_NORMALS, 3), "iid": (L.POST_UNIT, 3)}


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


class _Buffers:
    """Interval map of every tensor the programs can point into."""

    def __init__(self):
        self.items = []   # [start, end, kind, tensor]

    def add(self, t, kind):
        n = t.numel() * t.element_size()
        if n == 0:
            return
        assert t.is_contiguous(), "model image: non-contiguous buffer"
        self.items.append([t.data_ptr(), t.data_ptr() + n, kind, t])

    def finalize(self):
        # merge overlapping / duplicate intervals (views of one storage); data beats zero beats scratch
        self.items.sort(key=lambda it: (it[0], -it[1]))
        merged = []
        for it in self.items:
            if merged and it[0] < merged[-1][1]:
                m = merged[-1]
                assert it[1] <= m[1], "model image: partially overlapping buffers"
                m[2] = max(m[2], it[2])
                continue
            merged.append(list(it))
        self.items = merged
        self.starts = [it[0] for it in merged]

    def find(self, ptr):
        import bisect
        k = bisect.bisect_right(self.starts, ptr) - 1
        if k < 0 or ptr >= self.items[k][1]:
            return None
        return k, ptr - self.items[k][0]


def _pad(f, align=64):
    pos = f.tell()
    f.write(b"\0" * ((-pos) % align))
    return f.tell()


def export_model_image(pipe, path, *, ensemble_size, height, width, denoising_steps=None):
    """Write the model image of ``pipe`` for images of ``height`` x ``width`` (the size fed to the VAE: after the pipeline's
    ``processing_res`` resize), ``ensemble_size`` members per call and ``denoising_steps`` scheduler steps.  Works with the
    modules on the GPU (``pipe.to("cuda")``) or, without one, in the host-only mode.  Returns a dict describing the image."""
    unet, vae = pipe.unet, pipe.vae
    assert isinstance(unet, UNet2DConditionModelHIP) and isinstance(vae, AutoencoderKLHIP)
    if unet.ws is None:
        unet.dry()
        vae.dry()
    kind = pipe._kind
    post, cpred = _PRED[kind]
    B, T = int(ensemble_size), int(denoising_steps or pipe.default_denoising_steps)
    if pipe.empty_text_embed is None:
        pipe.encode_empty_text()
    if getattr(unet, "f16", False) or getattr(vae, "f16", False):
        raise ValueError("export_model_image: model images carry bf16 operands (the C host loads libmarigold_hip.so); "
                         "build the pipeline with compute_dtype=torch.bfloat16")
    unet.set_context(pipe.empty_text_embed)
    enc_seq, enc_in, enc_out = vae._program("encode", 1, height, width)
    h, w = enc_out.shape[-2:]
    den = unet.denoise_program(B, h, w, pipe.scheduler, T, rgb_broadcast=True)
    n_mod = getattr(pipe, "n_targets", 1)
    dec_seq, dec_in, dec_out = vae._program("decode", B * n_mod, h, w, post)

    bufs = _Buffers()
    for mod in (unet, vae):
        for v in mod.ws.cache.values():
            for t in _tensors(v):
                bufs.add(t, KIND_DATA)
        for t in mod.pool.all:
            bufs.add(t, KIND_SCRATCH)
    progs = [("vae.encode", enc_seq, {"rgb": enc_in, "latent": enc_out}),
             ("denoise", den.seq, dict({"rgb_latent": den.rgb_latent, "x": den.x}, **{f"noise{k}": nz for k, nz in enumerate(den.noises)})),
             ("vae.decode", dec_seq, {"latent": dec_in, "pred": dec_out})]
    for _, seq, io in progs:
        for t in seq.keep:
            # zero-initialised state the kernels rely on (V^T pad columns, tickets, flash workspace): tagged by the builder
            # (Builder.zeros_persistent -> OpSeq.zero_state); everything else held by the program is a constant or an input /
            # output slot
            bufs.add(t, KIND_ZERO if t.data_ptr() in seq.zero_state else KIND_DATA)
    bufs.finalize()
    # the workspace pools also hold buffers of programs built earlier in this process for OTHER shapes: only what these three
    # programs point into travels
    used = set()
    for _, seq, io in progs:
        for op in seq.ops:
            for s_ in range(16):
                if op.p[s_]:
                    hit = bufs.find(op.p[s_])
                    if hit is not None:
                        used.add(hit[0])
            if op.kind == L.OP_IGEMM and (op.i[29] or op.i[30]):
                hit = bufs.find((op.i[29] & 0xffffffff) | ((op.i[30] & 0xffffffff) << 32))
                if hit is not None:
                    used.add(hit[0])
        for t in io.values():
            hit = bufs.find(t.data_ptr())
            if hit is not None:
                used.add(hit[0])
    bufs.items = [it for k, it in enumerate(bufs.items) if it[2] != KIND_SCRATCH or k in used]
    bufs.starts = [it[0] for it in bufs.items]

    sz_op = ctypes.sizeof(L.MgOp)
    with open(path, "wb") as f:
        ptab, relocs_all, ops_all = [], [], []
        for name, seq, io in progs:
            relocs, raw = [], bytearray()
            for k, op in enumerate(seq.ops):
                c = L.MgOp()
                ctypes.memmove(ctypes.addressof(c), ctypes.addressof(op), sz_op)
                for s in range(16):
                    ptr = c.p[s]
                    if ptr:
                        hit = bufs.find(ptr)
                        if hit is None:
                            raise RuntimeError(f"model image: op {k} ({seq.labels[k]}) of {name} points outside every known buffer (p[{s}])")
                        relocs.append((k, s, hit[0], hit[1]))
                        c.p[s] = None
                if c.kind == L.OP_IGEMM and (c.i[29] or c.i[30]):
                    ptr = (c.i[29] & 0xffffffff) | ((c.i[30] & 0xffffffff) << 32)
                    hit = bufs.find(ptr)
                    if hit is None:
                        raise RuntimeError(f"model image: the row-statistics tickets of op {k} of {name} are not in a known buffer")
                    relocs.append((k, SLOT_LN_COUNTERS, hit[0], hit[1]))
                    c.i[29] = c.i[30] = 0
                raw += bytes(c)
            slots = []
            for nm, t in io.items():
                hit = bufs.find(t.data_ptr())
                slots.append((nm, hit[0], hit[1], t.numel() * t.element_size()))
            ptab.append((name, len(seq.ops), slots))
            relocs_all.append(relocs)
            ops_all.append(bytes(raw))
        # ---- layout: header, buffer table, program table, then 64-byte aligned blobs
        n_buf, n_prog = len(bufs.items), len(progs)
        HDR, BUF, PROG, SLOT, REL = "<8sIIII16I", "<QQII", "<32sIIQQI", "<24sIIQQ", "<IIIIQ"
        MAXSLOT = 16
        prog_sz = struct.calcsize(PROG) + 4 + MAXSLOT * struct.calcsize(SLOT)
        table_end = struct.calcsize(HDR) + n_buf * struct.calcsize(BUF) + n_prog * prog_sz
        f.write(b"\0" * table_end)
        buf_off = []
        for (start, end, k, t) in bufs.items:
            if k == KIND_DATA:
                off = _pad(f)
                f.write(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes() if t.numel() * t.element_size() == end - start
                        else bytes((ctypes.c_char * (end - start)).from_address(start)))
                buf_off.append(off)
            else:
                buf_off.append(0)
        ops_off, rel_off = [], []
        for raw, relocs in zip(ops_all, relocs_all):
            ops_off.append(_pad(f))
            f.write(raw)
            rel_off.append(_pad(f))
            for (k, s, b, o) in relocs:
                f.write(struct.pack(REL, k, s, b, 0, o))
        total = f.tell()
        f.seek(0)
        cfg = [B, height, width, h, w, T, cpred * n_mod, post, len(den.noises), sz_op, n_mod, dec_out.shape[-2], dec_out.shape[-1]] + [0] * 3
        f.write(struct.pack(HDR, MAGIC, VERSION, L.ABI_VERSION, n_buf, n_prog, *cfg))
        for (start, end, k, _t), off in zip(bufs.items, buf_off):
            f.write(struct.pack(BUF, end - start, off, k, 0))
        for (name, n_ops, slots), oo, ro, relocs in zip(ptab, ops_off, rel_off, relocs_all):
            assert len(slots) <= MAXSLOT, "model image: too many program slots"
            f.write(struct.pack(PROG, name.encode(), n_ops, len(relocs), oo, ro, len(slots)))
            f.write(b"\0" * 4)
            for (nm, b, o, n) in slots:
                f.write(struct.pack(SLOT, nm.encode(), b, 0, o, n))
            f.write(b"\0" * ((MAXSLOT - len(slots)) * struct.calcsize(SLOT)))
        assert f.tell() == table_end
    nbytes = {k: sum(it[1] - it[0] for it in bufs.items if it[2] == k) for k in (KIND_SCRATCH, KIND_ZERO, KIND_DATA)}
    return dict(path=path, file_bytes=total, buffers=n_buf, scratch_bytes=nbytes[KIND_SCRATCH], zero_bytes=nbytes[KIND_ZERO],
                data_bytes=nbytes[KIND_DATA], ops={n: c for (n, c, _s) in ptab}, latent_hw=(int(h), int(w)), B=B, steps=T,
                pred_channels=cpred * n_mod, step_noises=len(den.noises))


class ModelImage:
    """ctypes handle on a loaded model image - what a non-Python host does, spelled in Python for the tests and as a second,
    graph-free way to run a fixed-shape deployment: ``encode`` / ``denoise`` / ``decode`` take and return torch CUDA tensors
    but only their raw pointers cross into the library."""

    def __init__(self, path, device=0):
        lib = L.load()
        self._lib = lib
        self.handle = lib.mg_model_load(path.encode(), int(device))
        if not self.handle:
            L.check(1, "mg_model_load")
        cfg = (ctypes.c_int * 16)()
        L.check(lib.mg_model_info(self.handle, cfg), "mg_model_info")
        (self.B, self.H, self.W, self.h, self.w, self.steps, self.pred_channels, self.post, self.n_noise) = list(cfg)[:9]
        self.Hout, self.Wout = cfg[11], cfg[12]
        self.device = device

    def validate(self):
        L.check(self._lib.mg_model_validate(self.handle), "mg_model_validate")

    def encode(self, rgb):
        out = torch.empty(1, 4, self.h, self.w, device=rgb.device, dtype=torch.float32)
        rgb = rgb.to(torch.float32).contiguous()
        assert tuple(rgb.shape) == (1, 3, self.H, self.W)
        L.check(self._lib.mg_model_vae_encode(self.handle, rgb.data_ptr(), out.data_ptr(), O.current_stream_handle()), "mg_model_vae_encode")
        return out

    def denoise(self, rgb_latent, x, step_noises=None):
        x = x.to(torch.float32).contiguous().clone()
        nz = None
        if self.n_noise:
            nz = torch.stack(list(step_noises)).to(torch.float32).contiguous()
            assert nz.shape[0] == self.n_noise
        L.check(self._lib.mg_model_denoise(self.handle, rgb_latent.contiguous().data_ptr(), x.data_ptr(),
                                           None if nz is None else nz.data_ptr(), O.current_stream_handle()), "mg_model_denoise")
        return x

    def decode(self, latent):
        out = torch.empty(self.B, self.pred_channels, self.Hout, self.Wout, device=latent.device, dtype=torch.float32)
        L.check(self._lib.mg_model_vae_decode(self.handle, latent.to(torch.float32).contiguous().data_ptr(), out.data_ptr(),
                                              O.current_stream_handle()), "mg_model_vae_decode")
        return out

    def close(self):
        if self.handle:
            self._lib.mg_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
