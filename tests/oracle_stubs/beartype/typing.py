from typing import *  # noqa: F401,F403
from typing import List, Tuple, Dict, Iterable, Optional, Literal, Callable, Union, Any  # noqa: F401
