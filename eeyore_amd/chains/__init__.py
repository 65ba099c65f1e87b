from .storage import Chain, ChainFile, ChainList, ChainLists
from .chain_buffer import ChainBuffer
