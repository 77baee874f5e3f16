"""Fusion ablation (concat / minimal / weighted), 2-class drivable-area segmentation -- same entry
point as the reference's train_with_fusion_ablation.py, running on the MI355X-native path.

Environment knobs (defaults reproduce the reference's literals, train_with_fusion_ablation.py:17-61):
  KD_DATA_ROOT   PandaSet root (reference: a hard-coded Windows path)
  KD_TEACHER     optional checkpoint of a concat-fusion teacher: switches every variant to KD training
  KD_EPOCHS / KD_BATCH_SIZE   20 / 4
Launch with `python -m torch.distributed.run --nproc-per-node N` for data-parallel training: one process per GPU,
frames sharded over ranks in equal counts (every rank runs the same number of steps), rank 0's initial weights
broadcast, gradients all-reduced in buckets during backward (CE and KD training alike), BatchNorm statistics per
rank while training and rank 0's for validation, metrics summed over ranks, files written by rank 0 only.
"""
import json
import os

import torch
import torch.distributed as dist

from src.data_loading.pandaset_dataset import create_pandaset_dataloaders
from src.models.camera_encoder import TwinLiteEncoder
from src.models.fusion_module import CompleteSegmentationModel
from src.models.lidar_encoder import LiDAREncoder
from src.training.trainer import KDTrainer, Trainer

_MAIN = [True]


def log(*a, **k):
    """The script's own console output: one console under torchrun (rank 0); other ranks keep print() for diagnostics."""
    if _MAIN[0]:
        print(*a, **k)


def build_model(fusion_type, fusion_out_channels, device, num_classes=2):
    cam_enc = TwinLiteEncoder(return_multiscale=True)
    lidar_enc = LiDAREncoder(encoder_type="spatial", grid_size=(64, 64), use_vectorized=True)
    return CompleteSegmentationModel(camera_encoder=cam_enc, lidar_encoder=lidar_enc, num_classes=num_classes,
                                     fusion_type=fusion_type, fusion_out_channels=fusion_out_channels,
                                     camera_fpn_stages=["stage3", "stage4", "stage5"], camera_fpn_channels=128,
                                     output_mode="same").to(device)


def train_fusion_variant(fusion_type, fusion_out_channels, root, train_scenes, val_scenes, device):
    log(f"\n{'='*80}\nTRAINING: {fusion_type.upper()} FUSION\n{'='*80}")
    train_loader, val_loader = create_pandaset_dataloaders(
        root=root, train_scenes=train_scenes, val_scenes=val_scenes,
        batch_size=int(os.environ.get("KD_BATCH_SIZE", 4)), num_workers=2, verbose=False)
    model = build_model(fusion_type, fusion_out_channels, device)
    summary = model.get_architecture_summary()
    log(f"\nModel: {fusion_type}\n  Total params: {summary['total_params']}\n  Fusion params: {summary['fusion_params']}")
    kw = dict(lr=1e-3, weight_decay=1e-3, save_dir=f"checkpoints/fusion_ablation_{fusion_type}",
              class_weights=[0.4, 3.5], num_epochs=int(os.environ.get("KD_EPOCHS", 20)))
    teacher_ckpt = os.environ.get("KD_TEACHER")
    if teacher_ckpt:
        teacher = build_model("concat", 256, device)
        teacher.load_state_dict(torch.load(teacher_ckpt, map_location=device)["model_state"])
        trainer = KDTrainer(model, teacher, train_loader, val_loader, device, **kw)
    else:
        trainer = Trainer(model, train_loader, val_loader, device, **kw)
    best_miou = trainer.train()
    return best_miou, summary["total_params"], summary["fusion_params"]


def main():
    root = os.environ.get("KD_DATA_ROOT", r"D:\kelvin\Dataset\data")
    all_scenes = sorted([d for d in os.listdir(root) if d.isdigit()])
    n_train = int(0.8 * len(all_scenes))
    train_scenes, val_scenes = all_scenes[:n_train], all_scenes[n_train:]
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", 1)) > 1:
        # KD_REHEARSE_ON_ONE_GPU=1: all ranks on cuda:0 over gloo -- exercises the multi-rank code path on a 1-GPU box
        rehearse = os.environ.get("KD_REHEARSE_ON_ONE_GPU") == "1"
        torch.cuda.set_device(0 if rehearse else int(os.environ.get("LOCAL_RANK", 0)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")                       # the loaders shard the frames over ranks themselves
        else:                                                     # "nccl" is RCCL on ROCm; bind the communicator to this rank's GPU
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    main_rank = not dist.is_initialized() or dist.get_rank() == 0
    _MAIN[0] = main_rank
    log(f"\n{'='*80}\nFUSION ABLATION STUDY - 2-CLASS DRIVABLE AREA SEGMENTATION\n{'='*80}")
    log(f"Device: {device}\nScenes: {len(train_scenes)} train, {len(val_scenes)} val\n{'='*80}\n")
    results = {}
    for fusion_type, out_ch in (("concat", 256), ("minimal", 128), ("weighted", 128)):
        miou, total_params, fusion_params = train_fusion_variant(fusion_type, out_ch, root, train_scenes, val_scenes, device)
        results[fusion_type] = {"miou": miou, "total_params": total_params, "fusion_params": fusion_params}
    log(f"\n{'='*80}\nFUSION ABLATION RESULTS\n{'='*80}")
    log(f"{'Fusion':<12} {'mIoU':>8} {'Total Params':>15} {'Fusion Params':>15}\n" + "-" * 80)
    for ftype, data in results.items():
        log(f"{ftype:<12} {data['miou']:>8.4f} {data['total_params']:>15} {data['fusion_params']:>15}")
    best = max(results.items(), key=lambda x: x[1]["miou"])
    log(f"\n{'='*80}\nBEST FUSION: {best[0].upper()}\n  mIoU: {best[1]['miou']:.4f}\n  Total params: {best[1]['total_params']}\n{'='*80}\n")
    if main_rank:
        with open("fusion_ablation_results.json", "w") as f:
            json.dump(results, f, indent=2)
        log("Results saved to fusion_ablation_results.json")
    if dist.is_initialized():
        dist.barrier(device_ids=[torch.cuda.current_device()]) if dist.get_backend() == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
