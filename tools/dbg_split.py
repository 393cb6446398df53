import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import _libs as L
import motioncam_decoder_amd as M
dev = torch.device("cuda:0")
ctx = M.Context(0)
for (w, h, nb, dist, sig) in ((4032, 3024, 12, 1, 12.0), (4032, 3024, 12, 0, 0.0), (3840, 2160, 12, 1, 12.0), (7680, 4320, 12, 1, 12.0)):
    img = L.synth_image(w, h, nb, dist, sig, 4242)
    buf = L.encode7(img)
    hdr = np.frombuffer(buf[:16].tobytes(), np.uint32)
    ti = torch.from_numpy(buf).to(dev)
    for sp in (1, 2, 4):
        os.environ["MCRAW_SIDE_SPLIT"] = str(sp)
        to = torch.zeros(w * h * 2, dtype=torch.uint8, device=dev)
        fr = M.Context.make_frames([(ti.data_ptr(), ti.numel(), w, h, 7, to.data_ptr(), w * h)])
        wr, st = ctx.decode_batch(fr)
        got = to.cpu().numpy().view(np.uint16).reshape(h, w)
        bad = np.argwhere(got != img)
        rows = np.unique(bad[:, 0]) if len(bad) else []
        print(w, h, "dist", dist, "split", sp, "len", buf.size, "bitsOff", hdr[2], "refsOff", hdr[3], "status", st[0], "bad px", len(bad),
              "rows", (rows[0], rows[-1]) if len(rows) else None, "first", bad[0] if len(bad) else None,
              "diff", (int(got[tuple(bad[0])]), int(img[tuple(bad[0])])) if len(bad) else None)
