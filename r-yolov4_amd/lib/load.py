"""Drop-in for the reference's lib/load.py:9-21: `load_data(...) -> (dataset, dataloader)`.  The dataloader is a DeviceLoader: batches
are assembled on the MI355X from the resident image cache (datasets/base_dataset.py) instead of by 8 cv2 worker processes; it yields
the same (paths, imgs, targets) triples.  `custom` is the reference's third dataset type, whose class no longer matches its own base
class's constructor (datasets/custom_dataset.py:10-12 calls BaseDataset with five unrelated arguments) — it cannot be instantiated in
the reference either and raises NotImplementedError here.

Device-side extras (keyword only, all optional): `device`, `imread`, `imsize`, `pool_budget_bytes` / `pool_slab_bytes` (HBM budget of the
decoded-image pool, LRU beyond it), `decode_workers`, `side_stream` (assemble on a stream of the loader's own: the next batch is prepared
under the current training step), `rank` / `world_size` (data parallel: each rank iterates and pools its shard of the files)."""
from ..datasets.base_dataset import DeviceLoader
from ..datasets.DOTA_dataset import DOTADataset
from ..datasets.UCASAOD_dataset import UCASAODDataset


def load_data(data_dir, class_names, dataset_type, hyp, csl, img_size=608, batch_size=4, augment=False, shuffle=True, side_stream=True, rank=0,
              world_size=1, **device_kw):
    if dataset_type == "UCAS_AOD":
        dataset = UCASAODDataset(data_dir, class_names, hyp, img_size=img_size, augment=augment, csl=csl, **device_kw)
    elif dataset_type == "DOTA":
        dataset = DOTADataset(data_dir, class_names, hyp, img_size=img_size, augment=augment, csl=csl, **device_kw)
    else:
        raise NotImplementedError
    return dataset, DeviceLoader(dataset, batch_size, shuffle, side_stream=side_stream, rank=rank, world_size=world_size)
