from .base_model import BaseModel, LlamaModel, MistralModel, NeuralChatModel  # noqa: F401
