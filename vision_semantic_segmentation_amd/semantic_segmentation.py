"""SemanticSegmentation -- the reference's inference wrapper (src/semantic_segmentation.py:20-57)
on the HIP conv stack.

    seg = SemanticSegmentation(cfg.VISION_SEM_SEG.SEM_SEG_NETWORK)
    labels = seg.segmentation(image_rgb_u8)        # int64 ndarray [h/4-4, w/4-4], as the reference returns

Differences by design: weights come from a LOCAL checkpoint (``MODEL.WEIGHT``) in the reference's
format or, when that is empty, from a seeded random init -- the reference's
``resnext50_32x4d(pretrained=True)`` URL fetch (backbone/build.py:20) is never attempted;
normalisation (ToTensor + Normalize, :35-39) is fused into the stem kernel; the arg-max runs on the
GPU and ``segmentation_device`` hands back the uint8 label map without leaving HBM.
"""
import numpy as np
import torch

from . import _lib
from .network import SegNet, check_state_dict, load_checkpoint, random_state_dict


def _strict_bool(v, what):
    """True / False, or the strings yacs' merge_from_list may hand over; anything else is an error (bool('off') is True)."""
    if isinstance(v, bool):
        return v
    if isinstance(v, str) and v.strip().lower() in ("true", "on", "1", "yes"):
        return True
    if isinstance(v, str) and v.strip().lower() in ("false", "off", "0", "no"):
        return False
    raise ValueError("%s must be 'auto', True or False, not %r" % (what, v))


class SemanticSegmentation(object):
    def __init__(self, cfg, device=None, state_dict=None):
        """cfg: network configuration (cfg.VISION_SEM_SEG.SEM_SEG_NETWORK of base_cfg.py:96-112)."""
        _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("SemanticSegmentation needs a GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if cfg.MODEL.TYPE != "DeepLabv3+" or cfg.MODEL.BACKBONE != "resnext50_32x4d" or cfg.MODEL.OUTPUT_STRIDE not in (8, 16):
            raise NotImplementedError("only the reference configuration (DeepLabv3+, resnext50_32x4d; output stride 8, or 16) is built")
        self.output_stride = int(cfg.MODEL.OUTPUT_STRIDE)
        self.cfg = cfg
        self.num_classes = cfg.DATASET.NUM_CLASSES
        self.precision = getattr(cfg.MODEL, "PRECISION", "mixed")
        kw = dict(num_classes=self.num_classes, in_channels=cfg.DATASET.IN_CHANNELS, aspp_out=cfg.MODEL.ASPP.OUT_CHANNELS,
                  atrous_channels=tuple(cfg.MODEL.ASPP.ATROUS_CHANNELS), low_level_out=cfg.MODEL.DECODER.LOW_LEVEL_OUT_CHANNELS,
                  refine_channels=tuple(cfg.MODEL.DECODER.REFINE_CHANNELS))
        if state_dict is not None:
            self.state = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        elif cfg.MODEL.WEIGHT:
            self.state = load_checkpoint(cfg.MODEL.WEIGHT)                      # :31-32
        else:
            self.state = random_state_dict(seed=getattr(cfg.MODEL, "SEED", 0), **kw)
        check_state_dict(self.state, **kw)
        self._nets = {}
        # "mixed" self-check: the logits error of the mixed mode follows the WEIGHTS and was measured on seeded
        # draws only, so a real checkpoint is checked once, before its first plan, against the fp32-input HIP path (itself 2e-6 from
        # the fp32 reference) on several seeded frames, and every 16-bit tensor of the plan is scanned for Inf / NaN where it is
        # produced.  The check walks a LADDER of plans and keeps the first one that passes:
        #   "mixed"       f16 + FP4 matrix cores, layer1's first two blocks write one f16 plane (MIXED_LAYER1_LO = False starts here)
        #   "mixed+lo"    every layer1 block keeps its lo plane (the default configuration starts here)
        #   "split16"     the COMPLETE hi + lo pipeline (SegNet(full_split=True)): every tensor two f16 planes, every product three f16 passes, no
        #                 FP4 and no single-plane tensor anywhere -- what a calibrated (trained) checkpoint needs
        #   "f32"         fp32-input MFMA, the reference's precision (50 frames/s)
        sc = getattr(cfg.MODEL, "MIXED_SELF_CHECK", "auto")
        self._self_check = (state_dict is None and bool(cfg.MODEL.WEIGHT)) if sc == "auto" else _strict_bool(sc, "MODEL.MIXED_SELF_CHECK")
        self._layer1_lo = bool(getattr(cfg.MODEL, "MIXED_LAYER1_LO", True))
        self._rung = "mixed+lo" if self._layer1_lo else "mixed"     # which plan of the ladder the "mixed" precision currently means
        self.on_fail = str(getattr(cfg.MODEL, "MIXED_ON_FAIL", "f32"))
        if self.on_fail not in ("f32", "raise", "warn"):
            raise ValueError("MODEL.MIXED_ON_FAIL must be 'f32', 'raise' or 'warn', not %r" % self.on_fail)
        self.mixed_check = None            # {"size", "rel_err", "rung", "tried", ...} once the check has run

    LADDER = ("mixed", "mixed+lo", "split16", "f32")

    def net_for(self, h, w, raw_frame=None):
        """The compiled plan for an h x w network input (built on first use, kept per size).  raw_frame = (src_h, src_w): the plan
        that takes the raw BGR camera frame and pre-processes inside its first kernel."""
        key = (int(h), int(w)) if raw_frame is None else (int(h), int(w), int(raw_frame[0]), int(raw_frame[1]))
        if key not in self._nets:
            if self._self_check and self.precision == "mixed" and self.mixed_check is None:
                self.check_mixed_against_f32(key[0], key[1])
            rung = self._rung if self.precision == "mixed" else self.precision
            if rung == "f32" and raw_frame is not None:
                raise NotImplementedError("the self-check fell back to the fp32 plan, which has no pre-processing stem: "
                                          "use preprocess_device() + segmentation_device()")
            net = self._build(key[0], key[1], rung, raw_frame)
            if getattr(self.cfg.MODEL, "HIP_GRAPH", True):
                net.capture_graph()
            self._nets[key] = net
        return self._nets[key]

    def _build(self, h, w, rung, raw_frame=None):
        """rung: a plan of the ladder ("mixed", "mixed+lo", "split16") or a plain precision ("f32", "f16", "bf16")"""
        kw = dict(device=self.device, num_classes=self.num_classes, raw_frame=raw_frame, output_stride=self.output_stride)
        if rung in ("f32", "f16", "bf16"):
            return SegNet(self.state, h, w, precision=rung, **kw)
        if rung == "split16":
            return SegNet(self.state, h, w, precision="mixed", full_split=True, **kw)
        return SegNet(self.state, h, w, precision="mixed",
                      conv2_split=bool(getattr(self.cfg.MODEL, "MIXED_CONV2_SPLIT", True)),
                      gconv_mx=bool(getattr(self.cfg.MODEL, "MIXED_GCONV_MX", True)),
                      trunk_fp4=bool(getattr(self.cfg.MODEL, "MIXED_TRUNK_FP4", True)),
                      layer1_lo=(rung == "mixed+lo"), **kw)

    @staticmethod
    def decide_rung(tried, on_fail, n_frames=4):
        """The self-check's decision, apart from its measurements: `tried` = [{"rung", "rel_err", "finite", "nonfinite_ops", "passes"}] in
        ladder order -> (rung to use, its error, warning text or None).  The best PASSING rung wins; when none passes: on_fail "f32" ->
        the fp32 plan, "raise" -> RuntimeError, "warn" -> the best finite 16-bit plan (fp32 if there is none)."""
        ok = [t for t in tried if t["passes"]]
        if ok:
            t = min(ok, key=lambda t: t["rel_err"])
            return t["rung"], t["rel_err"], None
        summary = "; ".join("%s: %s" % (t["rung"], ("%.2e" % t["rel_err"]) if t["finite"] and not t["nonfinite_ops"]
                                        else "Inf/NaN in " + ", ".join(t["nonfinite_ops"] or ["the logits"])) for t in tried)
        msg = ("mixed precision: no 16-bit plan reproduces the fp32 logits of these weights within 1e-3 on %d frames (%s)" % (n_frames, summary))
        if on_fail == "raise":
            raise RuntimeError(msg)
        usable = [t for t in tried if t["finite"] and not t["nonfinite_ops"] and t["rel_err"] == t["rel_err"]]
        if on_fail == "warn" and usable:
            t = min(usable, key=lambda t: t["rel_err"])
            return t["rung"], t["rel_err"], msg + "; keeping '%s' (MODEL.MIXED_ON_FAIL = 'warn')" % t["rung"]
        return "f32", 0.0, msg + "; using the fp32 plan (4x slower)"

    @staticmethod
    def check_frames(h, w, seed=1, n_noise=3):
        """The self-check's frames: `n_noise` uniform-noise frames and one smooth frame (low-frequency colour ramps + a few blobs: real
        camera frames excite far fewer high-frequency channels than noise does)."""
        rng = np.random.default_rng(seed)
        frames = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(n_noise)]
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        smooth = np.stack([128 + 100 * np.sin(2 * np.pi * (xx / w * (1 + k) + yy / h * (2 - k) * 0.5) + k) for k in range(3)], axis=2)
        for _ in range(6):
            cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(0.05, 0.2) * min(h, w)
            smooth += (rng.uniform(-80, 80, size=3) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))[..., None])
        frames.append(np.clip(smooth, 0, 255).astype(np.uint8))
        return frames

    def check_mixed_against_f32(self, h, w, threshold=8e-4, seed=1):
        """Several seeded h x w frames (check_frames) through the plans of the ladder and through the fp32-input plan: the WORST
        max |dlogit| / max |logit| over the frames decides.  A plan FAILS when that figure is not finite, when its logits hold an Inf /
        NaN, when any of its tensors does (avl_seg_plan_nonfinite: an f16 overflow that a later ReLU would hide), or when the error is
        above 1e-3; it is ACCEPTED at once when the error is at most `threshold`, otherwise the next rung is measured too and the best
        passing one is kept.  When no 16-bit plan passes: MODEL.MIXED_ON_FAIL = "f32" (default) uses the fp32 plan from now on,
        "raise" raises, "warn" keeps the best-effort 16-bit plan with a warning.  A checkpoint whose fp32 logits are not finite raises.
        Returns (and keeps in .mixed_check) what was measured."""
        import warnings
        frames = [torch.from_numpy(f).to(self.device) for f in self.check_frames(h, w, seed)]
        ref = self._build(h, w, "f32")
        refs = []
        for f in frames:
            ref.forward(f)
            refs.append(ref.logits.float().clone())
        del ref
        if not all(bool(torch.isfinite(r).all()) for r in refs):
            raise RuntimeError("the checkpoint's logits are not finite even in fp32: %s" % (self.cfg.MODEL.WEIGHT or "<state_dict>"))
        tried = []
        start = self.LADDER.index(self._rung)
        for rung in self.LADDER[start:-1]:
            net = self._build(h, w, rung)
            err, finite = 0.0, True
            for f, r in zip(frames, refs):
                net.forward(f)
                lg = net.logits.float()
                finite = finite and bool(torch.isfinite(lg).all())
                e = float((lg - r).abs().max()) / float(r.abs().max())
                err = e if not (e <= err) else err          # NaN-safe maximum: a NaN error sticks
            bad = net.nonfinite_counts() if finite else {"logits": 1}
            del net
            ok = finite and not bad and (err <= 1e-3)
            tried.append({"rung": rung, "rel_err": err, "finite": finite, "nonfinite_ops": sorted(bad)[:4], "passes": ok})
            if ok and err <= threshold:
                break
        torch.cuda.empty_cache()
        self.mixed_check = {"size": (h, w), "threshold": threshold, "frames": len(frames), "tried": tried}
        self._rung, err, warning = self.decide_rung(tried, self.on_fail, len(frames))
        if warning:
            warnings.warn(warning)
        self._layer1_lo = self._rung != "mixed"
        self.mixed_check.update(rung=self._rung, layer1_lo=self._layer1_lo, rel_err=err)
        return self.mixed_check

    def segmentation_device(self, image_in):
        """uint8 RGB [h,w,3] (ndarray or CUDA tensor) -> uint8 CUDA tensor [h/4-4, w/4-4]."""
        h, w = int(image_in.shape[0]), int(image_in.shape[1])
        net = self.net_for(h, w)
        return net.forward(image_in)

    def segmentation_device_raw(self, bgr, K=None, dist=None, factor=1):
        """The node's chain from the camera frame on (vision_semantic_segmentation_node.py:83-102) in the network's own kernels:
        uint8 BGR [H,W,3] (ndarray or CUDA tensor) -> BGR->RGB, cv2.undistort(K, dist) (skipped when None), INTER_AREA by the integer
        `factor`, normalise, network, arg-max -> uint8 CUDA tensor.  Same labels as preprocess_device() + segmentation_device(), without
        the RGB frame in between."""
        if self.precision == "f32":
            raise NotImplementedError("the pre-processing stem is a 16-bit MFMA kernel; use preprocess_device() with PRECISION f32")
        H, W = int(bgr.shape[0]), int(bgr.shape[1])
        net = self.net_for(H // factor, W // factor, raw_frame=(H, W))
        net.set_camera(K, dist)
        return net.forward(bgr)

    def segmentation(self, image_in):
        """semantic_segmentation.py:41-57: numpy (h, w, 3) RGB -> int64 numpy label map."""
        labels = self.segmentation_device(image_in)
        return labels.cpu().numpy().astype(np.int64)

    def logits(self, image_in):
        """float32 CUDA tensor [K, h', w'] (the reference's layout) of model(x, upsample_pred=False)."""
        h, w = int(image_in.shape[0]), int(image_in.shape[1])
        net = self.net_for(h, w)
        net.forward(image_in)
        return net.logits.permute(2, 0, 1)
