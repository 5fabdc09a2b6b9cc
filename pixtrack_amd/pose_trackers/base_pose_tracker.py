"""Per-video loop: refine -> (relocalize on failure) -> update references
(reference pixtrack/pose_trackers/base_pose_tracker.py:5-37)."""
import numpy as np

try:
    import tqdm
except ImportError:  # pragma: no cover
    tqdm = None


class PoseTracker:
    def relocalize(self, query):
        raise NotImplementedError

    def refine(self, query):
        raise NotImplementedError

    def get_query_frame_iterator(self, query_path, max_frames):
        raise NotImplementedError

    def update_reference_ids(self):
        raise NotImplementedError

    def run_single_frame(self, frame):
        pose_success = self.refine(frame)
        if not pose_success:
            # the reference hands the whole (path, image) tuple to relocalize (Appendix D.4)
            self.relocalize(frame)
        self.update_reference_ids()

    def run(self, query_path, max_frames=np.inf):
        frame_iterator = self.get_query_frame_iterator(query_path, max_frames)
        self.pbar = tqdm.tqdm(frame_iterator) if tqdm is not None else frame_iterator
        for frame in self.pbar:
            self.run_single_frame(frame)
