from .chain import Chain
from .chain_file import ChainFile
from .chain_list import ChainList
from .chain_lists import ChainLists
from .chain_buffer import ChainBuffer
