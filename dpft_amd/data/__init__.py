"""Input pipeline at multi-GPU rates: sharded sampling, the reference's collate contract, pinned
prefetch with an upload stream, and the per-sample online transforms (radar scaling, camera resize) moved onto the GPU."""
from dpft_amd.data.loader import (BlockShardedSampler, PrefetchLoader, ShardedSampler, listed_collating, load_listed,   # noqa: F401
                                  load_listed_eval)
from dpft_amd.data.preprocess import GpuPreprocessor, resized_output_size                        # noqa: F401
from dpft_amd.data.synthetic_raw import SyntheticRawDataset                                      # noqa: F401
from dpft_amd.data.radar_projection import doppler_raster, radar_projection                      # noqa: F401
from dpft_amd.data.kradar import KRadarFolderDataset, initialize_kradar                       # noqa: F401
