"""Inference adapter -- counterpart of AutonomousDriver.preprocess_image / predict_controls
(reference model/autonomous_drive.py:481-485, 897-920).

The reference resizes the 800x600 camera frame with cv2.resize (INTER_LINEAR) on the host; cv2 is
not part of this engine, so frames are expected at the network resolution (88x200) already.  The
/255, HWC->CHW and Normalize(mean, std) steps are fused into one HIP kernel that writes the NHWC
tensor the stem convolution reads.
"""
from __future__ import annotations

import numpy as np
import torch

IMG_MEAN = (0.485, 0.456, 0.406)       # autonomous_drive.py:481
IMG_STD = (0.229, 0.224, 0.225)        # :482
IMG_WIDTH, IMG_HEIGHT = 200, 88        # :483-484
SPEED_NORM_FACTOR = 90.0               # :485


class Predictor:
    """Holds pinned staging buffers so a 20 Hz control loop does one H2D and one D2H copy per
    tick (the reference does four ``.item()`` syncs, :918-920)."""

    def __init__(self, model, batch=1, height=IMG_HEIGHT, width=IMG_WIDTH, use_graph=True):
        self.model = model.eval()
        self.eng = model.engine()
        dev = self.eng.device
        self.batch = batch
        self.use_graph = use_graph
        self.stream = torch.cuda.Stream(device=dev)      # hipGraph capture needs its own stream
        self.ctrl_dev = torch.empty(batch, 3, dtype=torch.float32, device=dev)
        self.spd_out_dev = torch.empty(batch, dtype=torch.float32, device=dev)
        self.out_dev = torch.empty(batch, 4, dtype=torch.float32, device=dev)
        self.frames_host = torch.empty(batch, height, width, 3, dtype=torch.uint8).pin_memory()
        self.frames_dev = torch.empty(batch, height, width, 3, dtype=torch.uint8, device=dev)
        self.speed_host = torch.empty(batch, dtype=torch.float32).pin_memory()
        self.cmd_host = torch.empty(batch, dtype=torch.int64).pin_memory()
        self.speed_dev = torch.empty(batch, dtype=torch.float32, device=dev)
        self.cmd_dev = torch.empty(batch, dtype=torch.int64, device=dev)
        self.out_host = torch.empty(batch, 4, dtype=torch.float32).pin_memory()

    @torch.no_grad()
    def predict_batch(self, frames_u8, speeds_kmh, commands):
        """frames uint8 [B,88,200,3] RGB, km/h, command idx -> np.float32 [B,4] =
        (steer, throttle, brake, speed_kmh)."""
        if self.model.engine() is not self.eng:
            self.__init__(self.model, self.batch, self.frames_host.size(1),
                          self.frames_host.size(2), self.use_graph)
        if self.model.training:
            self.model.eval()
        self.frames_host.copy_(torch.as_tensor(frames_u8))
        sp = np.minimum(np.asarray(speeds_kmh, dtype=np.float32) / np.float32(SPEED_NORM_FACTOR),
                        np.float32(1.0))
        self.speed_host.copy_(torch.from_numpy(sp))
        self.cmd_host.copy_(torch.as_tensor(commands, dtype=torch.int64))
        with torch.cuda.stream(self.stream):
            self.frames_dev.copy_(self.frames_host, non_blocking=True)
            self.speed_dev.copy_(self.speed_host, non_blocking=True)
            self.cmd_dev.copy_(self.cmd_host, non_blocking=True)
            self.eng.run_forward_u8(self.frames_dev, self.speed_dev, self.cmd_dev,
                                    out=(self.ctrl_dev, self.spd_out_dev), graph=self.use_graph)
            self.out_dev[:, :3].copy_(self.ctrl_dev)
            torch.mul(self.spd_out_dev, SPEED_NORM_FACTOR, out=self.out_dev[:, 3])
            self.out_host.copy_(self.out_dev, non_blocking=True)
            self.stream.synchronize()
        return self.out_host.numpy().copy()

    def predict_controls(self, image_rgb_u8, speed_kmh, command_idx):
        """Same return tuple as the reference's predict_controls (:918-920)."""
        r = self.predict_batch(np.asarray(image_rgb_u8)[None], [speed_kmh], [command_idx])[0]
        return float(r[0]), float(r[1]), float(r[2]), float(r[3])
