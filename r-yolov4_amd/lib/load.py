"""Drop-in for the reference's lib/load.py:9-21: `load_data(...) -> (dataset, dataloader)`.  The dataloader is a DeviceLoader: batches
are assembled on the MI355X from the resident image cache (datasets/base_dataset.py) instead of by 8 cv2 worker processes; it yields
the same (paths, imgs, targets) triples.  `custom` is the reference's third dataset type, whose class no longer matches its own base
class's constructor (datasets/custom_dataset.py:10-12 calls BaseDataset with five unrelated arguments) — it cannot be instantiated in
the reference either and raises NotImplementedError here."""
from ..datasets.base_dataset import DeviceLoader
from ..datasets.DOTA_dataset import DOTADataset
from ..datasets.UCASAOD_dataset import UCASAODDataset


def load_data(data_dir, class_names, dataset_type, hyp, csl, img_size=608, batch_size=4, augment=False, shuffle=True, **device_kw):
    if dataset_type == "UCAS_AOD":
        dataset = UCASAODDataset(data_dir, class_names, hyp, img_size=img_size, augment=augment, csl=csl, **device_kw)
    elif dataset_type == "DOTA":
        dataset = DOTADataset(data_dir, class_names, hyp, img_size=img_size, augment=augment, csl=csl, **device_kw)
    else:
        raise NotImplementedError
    return dataset, DeviceLoader(dataset, batch_size, shuffle)
