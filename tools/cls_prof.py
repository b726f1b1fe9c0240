import torch
from radialog_amd import synth
from radialog_amd.chexpert_model import ChexpertClassifier
from radialog_amd.engine import synth_getter
m = ChexpertClassifier(num_classes=14, dtype="f16", max_batch=8)
m.set_weight_getter(synth_getter(m.cfg, torch.device("cuda", 0), lora=False))
x = synth.synth_images(1, 488, seed=3).cuda()
for _ in range(3): out = m(x)
torch.cuda.synchronize()
