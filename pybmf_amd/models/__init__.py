from .BaseModel import BaseModel
from .ContinuousModel import ContinuousModel
from .BinaryMFPenalty import BinaryMFPenalty
from .WNMF import WNMF
from .PNLPF import PNLPF
from .BinaryMFThreshold import BinaryMFThreshold
from .ELBMF import ELBMF
from .PRIMP import PRIMP

__all__ = ["BaseModel", "ContinuousModel", "BinaryMFPenalty", "PNLPF", "WNMF", "BinaryMFThreshold", "ELBMF", "PRIMP"]
