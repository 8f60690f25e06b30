"""FusedFeatureService — the three services as ONE subscriber of `video.preprocessed`: the clip is
decoded once, and the three result files / subjects are produced in the order yolo -> sam3 -> dinov3 (ml-pipeline
triggers on `pipeline.dinov3` and expects the others on disk).  Running YOLO first also removes the reference's race
between the yolo and sam3 services (Appendix C-4): SAM3 always finds the YOLO file."""


class FusedFeatureService:
    def __init__(self, yolo_pipeline, sam3_pipeline, dinov3_pipeline):
        self.yolo, self.sam3, self.dinov3 = yolo_pipeline, sam3_pipeline, dinov3_pipeline
        self.nats_client = yolo_pipeline.nats_client
        self.config = yolo_pipeline.config

    async def process_video(self, video_data):
        await self.yolo.process_video(video_data)
        await self.sam3.process_video(video_data)
        await self.dinov3.process_video(video_data)

    async def start(self):
        await self.nats_client.connect()
        await self.nats_client.subscribe(self.config["nats"]["subjects"]["video_preprocessed"], self.process_video)
