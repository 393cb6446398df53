"""Helpers for the -m gpu tests: drive the C ABI with HBM-resident buffers."""
import numpy as np
import torch

import motioncam_decoder_amd as M


def decode_batch_device(ctx, items, fill=0xA5, out_rows_extra=0):
    """items: list of (type, w, h, buf[np.uint8]).  Inputs are uploaded to HBM first;
    returns (written, status, outs[list of np.uint16 (h, w)])."""
    dev = torch.device("cuda:0")
    ins, outs, descs = [], [], []
    for typ, w, h, buf in items:
        t_in = torch.from_numpy(np.ascontiguousarray(buf)).to(dev)
        t_out = torch.full((max(w, 0) * max(h + out_rows_extra, 0) * 2 + 16,), fill, dtype=torch.uint8, device=dev)
        ins.append(t_in)
        outs.append(t_out)
        descs.append((t_in.data_ptr(), t_in.numel(), w, h, typ, t_out.data_ptr(), max(w, 0) * max(h, 0)))
    torch.cuda.synchronize()
    frames = M.Context.make_frames(descs)
    written, status = ctx.decode_batch(frames, mem=M.MEM_DEVICE, stream=None, want_status=True)
    torch.cuda.synchronize()
    res = []
    for (typ, w, h, buf), t_out in zip(items, outs):
        a = t_out.cpu().numpy()[: w * (h + out_rows_extra) * 2].view(np.uint16).reshape(h + out_rows_extra, w)
        res.append(a)
    return written, status, res
