"""`rtwm`: drop-in import name of the reference package, backed by echoseal_amd (MI355X)."""
from echoseal_amd.embedder import WatermarkEmbedder
from echoseal_amd.detector import WatermarkDetector

__all__ = ["WatermarkEmbedder", "WatermarkDetector"]
