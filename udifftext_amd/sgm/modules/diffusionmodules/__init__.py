from .denoiser import Denoiser
from .discretizer import Discretization
from .loss import StandardDiffusionLoss
from .model import Decoder, Encoder
from .openaimodel import UnifiedUNetModel
from .sampling import BaseDiffusionSampler
from .wrappers import OpenAIWrapper
