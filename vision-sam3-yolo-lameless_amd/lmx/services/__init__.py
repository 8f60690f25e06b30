"""Host-side mirror of the three hot-path services of the reference: same NATS subjects, same
on-disk JSON schema (Appendix B), same sampling and quirks (Appendix C), with the third-party model call replaced by
the liblmx path.  ``YOLOPipeline`` / ``SAM3Pipeline`` / ``DINOv3Pipeline`` keep the reference's class and method names
(services/*/app/main.py); ``FusedFeatureService`` is the single-process, single-decode composition of the three; ``TrackingService`` is the consumer
directly after them (ByteTrack over `pipeline.yolo`, Re-ID over `pipeline.dinov3`; services/tracking-service)."""
from .dinov3_pipeline import DINOv3Pipeline  # noqa: F401
from .fused import FusedFeatureService  # noqa: F401
from .sam3_pipeline import SAM3Pipeline  # noqa: F401
from .yolo_pipeline import YOLOPipeline  # noqa: F401
from .pose import PoseEstimator  # noqa: F401
from .curation import CowTracker  # noqa: F401
from .tracking import TrackingService  # noqa: F401
