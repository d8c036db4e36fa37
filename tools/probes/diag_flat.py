import os, sys, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dcnet_amd import losses, ops
from dcnet_amd.model import grounding_model
from dcnet_amd.parallel import FlatGradAllReduce, freeze_gradless
from dcnet_amd.train import make_optimizer
from dcnet_amd.utils.synth import synth_boxes, synth_inputs
dev = torch.device("cuda:0")
size, n = 256, 2
def build():
    torch.manual_seed(1234)
    m = grounding_model(corpus=list(range(1000)), light=False, emb_size=512, coordmap=True, bert_model="x", dataset="vid", img_size=size,
                        config_path=os.path.join(ROOT, "model", "yolov3.cfg"), weights_path=None).to(dev)
    m.train(); freeze_gradless(m)
    return m
image, word_id, word_mask = synth_inputs(n, size, seed=100)
bbox = synth_boxes(n, size, seed=100)
image, word_id, word_mask, bbox = image.to(dev), word_id.to(dev), word_mask.to(dev), bbox.to(dev)
m = build(); opt = make_optimizer(m, 1e-4)
random.seed(13)
out = m(image, word_id, word_mask); loss, _ = losses.total_loss(out, bbox, size); loss.backward()
none = [k for k, p in m.named_parameters() if p.requires_grad and p.grad is None]
zero = [k for k, p in m.named_parameters() if p.requires_grad and p.grad is not None and float(p.grad.abs().max()) == 0.0]
print("requires_grad but grad None:", none)
print("grad all-zero:", zero)
for mode in ("unbound", "bound"):
    m = build(); opt = make_optimizer(m, 1e-4)
    flat = FlatGradAllReduce(m.parameters()).bind() if mode == "bound" else None
    random.seed(13)
    ls = []
    for it in range(4):
        out = m(image, word_id, word_mask); loss, _ = losses.total_loss(out, bbox, size)
        if flat is not None: flat.zero()
        else: opt.zero_grad(set_to_none=True)
        loss.backward(); opt.step(); ls.append(float(loss))
    print(mode, ls)
