"""GPU: TrainEngine (graph-captured passes) over the data-parallel path in a world of one rank
(UD_FORCE_COLLECTIVES=1: SyncBN exchange + streamed gradient buckets inside the captured graphs)."""
import copy, os, sys
os.environ.setdefault("UD_FORCE_COLLECTIVES", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from tests.test_train_engine import CONFIG
from unidefense_amd.engine import get_engine
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
cfg = copy.deepcopy(CONFIG)
cfg["config"]["num_steps"], cfg["config"]["log_steps"] = 4, 2
eng = get_engine("FE")(cfg, "Train")
print("wrapped:", type(eng.model).__name__, "graphs:", eng.use_graphs)
log = eng.train()
print("captured:", sorted(k for st in eng._graphs.values() for k in st if k in ("g1", "g2")), log["total_loss"])
dist.destroy_process_group()
