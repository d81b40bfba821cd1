"""Metrics stub.  The reference's MetricsManager (torchok/metrics/metrics_manager.py:78-206) wraps
torchmetrics / FAISS / ranx — epoch-end, host-side, third-party, outside the hot-path scope
(SURVEY.md §2 #30).  The Task API only needs ``update`` / ``on_epoch_end`` to exist; configured
metrics are recorded by name and otherwise ignored."""
from typing import Dict, List


class MetricsManager:
    def __init__(self, params: List[dict]):
        self.configured = [m.get('name') for m in (params or [])]

    def update(self, phase, *args, **kwargs) -> None:
        return None

    def on_epoch_end(self, phase) -> Dict[str, float]:
        return {}
