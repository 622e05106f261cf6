"""Trainable interface (reference: src/models/interfaces/trainable.py:8-22)."""
from abc import ABC, abstractmethod


class Trainable(ABC):
    @property
    @abstractmethod
    def objective(self):
        """The objective to MINIMISE, re-evaluated from the current parameters."""
        pass
