from .BaseModel import BaseModel
from .ContinuousModel import ContinuousModel
from .BinaryMFPenalty import BinaryMFPenalty
from .WNMF import WNMF
from .BinaryMFThreshold import BinaryMFThreshold

__all__ = ["BaseModel", "ContinuousModel", "BinaryMFPenalty", "WNMF", "BinaryMFThreshold"]
