"""Concat-fusion PandaSet training -- same entry point as the reference's train_pandaset.py
(3-class head, class weights [0.39, 2.61, 33.09], 30 epochs, interactive resume prompt:
train_pandaset.py:79-163), on the MI355X-native path.  KD_DATA_ROOT overrides the dataset root."""
import os

import torch

from src.data_loading.pandaset_dataset import create_pandaset_dataloaders
from src.models.camera_encoder import TwinLiteEncoder
from src.models.fusion_module import CompleteSegmentationModel
from src.models.lidar_encoder import LiDAREncoder
from src.training.trainer import Trainer


def main():
    root = os.environ.get("KD_DATA_ROOT", r"D:\kelvin\Dataset\data")
    all_scenes = sorted([d for d in os.listdir(root) if d.isdigit()])
    n_train = int(0.8 * len(all_scenes))
    train_scenes, val_scenes = all_scenes[:n_train], all_scenes[n_train:]
    print(f"Found {len(all_scenes)} scenes\nTrain: {len(train_scenes)} scenes | Val: {len(val_scenes)} scenes")
    train_loader, val_loader = create_pandaset_dataloaders(root=root, train_scenes=train_scenes, val_scenes=val_scenes,
                                                           batch_size=4, num_workers=2, verbose=True)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    print(f"\nUsing device: {device}\n\nBuilding model...")
    model = CompleteSegmentationModel(
        camera_encoder=TwinLiteEncoder(return_multiscale=True),
        lidar_encoder=LiDAREncoder(encoder_type="spatial", grid_size=(64, 64), use_vectorized=True),
        num_classes=3, fusion_type="concat", fusion_out_channels=256,
        camera_fpn_stages=["stage3", "stage4", "stage5"], camera_fpn_channels=128, output_mode="same").to(device)
    s = model.get_architecture_summary()
    print("\nModel Architecture:")
    for k in ("camera", "lidar", "fusion", "head", "total"):
        print(f"  {k.capitalize()} params: {s[k + '_params']}")
    save_dir = "checkpoints/pandaset_weighted"
    trainer = Trainer(model=model, train_loader=train_loader, val_loader=val_loader, device=device, lr=1e-3,
                      weight_decay=1e-3, save_dir=save_dir, class_weights=[0.39, 2.61, 33.09], num_epochs=30)
    start_epoch = 0
    ckpt_path = os.path.join(save_dir, "latest.pth")
    if os.path.exists(ckpt_path):
        response = input(f"\nFound checkpoint at {ckpt_path}. Resume training? (y/n): ")
        if response.lower() == "y":
            start_epoch = trainer.load_checkpoint(ckpt_path)
    trainer.train(start_epoch=start_epoch)


if __name__ == "__main__":
    main()
