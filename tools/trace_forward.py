import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import estts
from estts import configs, ops, weights
from estts.forward_transformer import ForwardTransformer
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_utterance
cfg = configs.forward_config()
w = weights.init_forward_transformer(cfg, seed=1234)
tokens, durs = synthetic_utterance(cfg, 1, 100, 600, seed=1234)
m = ForwardTransformer(w, cfg)
tok = torch.from_numpy(tokens).cuda(); dur = torch.from_numpy(durs.astype(np.float32)).cuda()
m.call(tok, target_durations=dur, max_frames=600)
ops.start_trace()
m.call(tok, target_durations=dur, max_frames=600)
tr = ops.stop_trace()
for i, (kind, info, _) in enumerate(tr):
    print(i, kind, {k: v for k, v in info.items() if k not in ('flops', 'bytes')})
